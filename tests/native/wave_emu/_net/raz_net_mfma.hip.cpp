// raz_net_mfma.hip — the narrow-net forward pass on the matrix cores: one wavefront per position,
// every 3x3 convolution an implicit GEMM  D[16 squares x 16 channels] += A[16 x 4] * B[4 x 16]
// on v_mfma_f32_16x16x4_f32 (f32 in, f32 accumulate; bit-identical to a k-ordered fmaf chain, so
// this kernel, k_net_wave and the CPU oracle agree to the last bit — raznet-forward-v1).
//
// Why MFMA although f32 MFMA has the same FLOP rate as f32 VALU: the VALU form needs one operand
// fetch per FMA (an LDS read per activation, a scalar load per 16 weights) and ran at 15 % of peak;
// an MFMA consumes ONE activation register and ONE weight register per 1024 MACs.  The weights of
// a whole layer (9*Cin/4 registers per 16-channel tile) sit in VGPRs, each k-step costs four
// ds_read_b32 (one per 16-square M tile) and four MFMAs on four independent accumulators.
//
// LDS: activations are kept as zero-haloed planes, 12 floats per board row with the 8 squares at
// columns 4..11 (plane index (y+1)*12 + x + 4, plane stride 136 floats): off-board taps read the
// halo (zero) so there is no predicate in the k-loop, every tap/channel variation is an immediate
// offset of one per-lane base address, and the D fragment (4 consecutive squares of one channel
// per lane) is written with one aligned ds_write_b128.  Two buffers: `a` (block input, updated in
// place by the residual add) and `t`.
#include <hip/hip_runtime.h>
#include "raz_bitboard.h"
#include "raz_detmath.h"
#include "raz_internal.h"
#include "raz_net_layout.h"
#include "raz_net_wave.h"   // plane layout, conv_layer, the in-wave forward

namespace {

// PROF: lane 0 records s_memtime ticks at phase boundaries into prof[pos][8] (debug launches only:
// RAZ_NET_PROF=1 + a scratch buffer).
#define RAZ_NET_TICK(k) \
    if (PROF && lane == 0) prof[(size_t)pos * 8 + (k)] = __builtin_amdgcn_s_memtime()

// Persistent: min(n, 2048) single-wave workgroups (8 per CU = the LDS limit), each looping over
// positions.  The LDS halos are zeroed once per workgroup; for the F = 16, R = 1 net (mini.yml) all
// 77 weight registers of the three conv layers stay resident across positions.
template <int F, bool PROF>
__global__ __launch_bounds__(64, F == 16 ? 2 : 1) void k_net_mfma(const float* __restrict__ W, int R, int V,
                                                 const raz_bb* __restrict__ own,
                                                 const raz_bb* __restrict__ enemy,
                                                 const uint8_t* __restrict__ active,
                                                 float* __restrict__ policy, float* __restrict__ value, int n,
                                                 unsigned long long* prof) {
    alignas(16) static float smem[RAZ_EMU_LDS_FLOATS];
    const int lane = threadIdx.x;
    constexpr int NBUF = 2;   // (a single in-place buffer was tried for F == 16: registers, not LDS, bound the occupancy)
    float* bufA = smem;
    float* bufT = smem + (NBUF - 1) * F * PS;
    float* head = smem + NBUF * F * PS;  // ph[128] vh[64] h1[V]
    {  // zero the buffers once: halos must read as 0, interiors are overwritten for every position
        f32x4* z = (f32x4*)smem;
        for (int j = lane; j < NBUF * F * PS / 4; j += 64) z[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    const float* H = W + heads_off(F, R);
    const float* pol_w = H;
    const float* pol_b = pol_w + 2 * F;
    const float* pfc_w = pol_b + 2;
    const float* pfc_b = pfc_w + 128 * 64;
    const float* val_w = pfc_b + 64;
    const float* val_b = val_w + F;
    const float* v1_w = val_b + 1;
    const float* v1_b = v1_w + 64 * V;
    const float* v2_w = v1_b + V;
    const float* v2_b = v2_w + V;
    float* ph = head;
    float* vh = head + 128;
    float* h1 = head + 192;
    constexpr bool HOIST = (F == 16);
    const bool pre = HOIST && R == 1;
    float w0[5], w1[HOIST ? 36 : 1], w2[HOIST ? 36 : 1], pb0 = 0.f, pb1 = 0.f, pb2 = 0.f;
    if (HOIST) {
        if (pre) {
            load_wregs<5>(W + mfma_layer_off(F, R, V, 0), 0, lane, w0);
            load_wregs<HOIST ? 36 : 1>(W + mfma_layer_off(F, R, V, 1), 0, lane, w1);
            load_wregs<HOIST ? 36 : 1>(W + mfma_layer_off(F, R, V, 2), 0, lane, w2);
            pb0 = (W + conv_off(F, 0) + (size_t)F * 9 * 2)[lane & 15];
            pb1 = (W + conv_off(F, 1) + (size_t)F * 9 * F)[lane & 15];
            pb2 = (W + conv_off(F, 2) + (size_t)F * 9 * F)[lane & 15];
        }
    }
    const float dummy5[5] = {0, 0, 0, 0, 0};
    for (int pos = blockIdx.x; pos < n; pos += gridDim.x) {
        if (active && !active[pos]) continue;
        RAZ_NET_TICK(0);
        const raz_bb bo = own[pos], be = enemy[pos];
        // the two input bit planes go to planes 0/1 of bufT (overwritten again by the first block conv)
        bufT[pidx(lane)] = (float)((bo >> lane) & 1ULL);
        bufT[PS + pidx(lane)] = (float)((be >> lane) & 1ULL);
        __syncthreads();
        RAZ_NET_TICK(1);
        if (HOIST && pre) {
            if constexpr (HOIST) {
                conv_layer<F, 2, true, false, true>(nullptr, nullptr, bufT, bufA, lane, w0, pb0);
                RAZ_NET_TICK(2);
                conv_layer<F, F, false, false, true>(nullptr, nullptr, bufA, bufT, lane, w1, pb1);
                conv_layer<F, F, false, true, true>(nullptr, nullptr, bufT, bufA, lane, w2, pb2);
            }
        } else {
            float dummyK[LayerK<F, F, false>::KS];
            conv_layer<F, 2, true, false, false>(W + mfma_layer_off(F, R, V, 0), W + conv_off(F, 0) + (size_t)F * 9 * 2, bufT,
                                                 bufA, lane, dummy5, 0.f);
            RAZ_NET_TICK(2);
            for (int r = 0; r < R; ++r) {
                const int l1 = 1 + 2 * r, l2 = 2 + 2 * r;
                conv_layer<F, F, false, false, false>(W + mfma_layer_off(F, R, V, l1), W + conv_off(F, l1) + (size_t)F * 9 * F,
                                                      bufA, bufT, lane, dummyK, 0.f);
                conv_layer<F, F, false, true, false>(W + mfma_layer_off(F, R, V, l2), W + conv_off(F, l2) + (size_t)F * 9 * F,
                                                     bufT, bufA, lane, dummyK, 0.f);
            }
        }
        RAZ_NET_TICK(3);
        {
            const float* a = bufA + pidx(lane);
            float p0 = pol_b[0], p1 = pol_b[1], v0 = val_b[0];
#pragma unroll 8
            for (int ic = 0; ic < F; ++ic) {
                const float xv = a[ic * PS];
                p0 = fmaf(xv, pol_w[ic], p0);
                p1 = fmaf(xv, pol_w[F + ic], p1);
                v0 = fmaf(xv, val_w[ic], v0);
            }
            ph[lane] = p0 > 0.0f ? p0 : 0.0f;
            ph[64 + lane] = p1 > 0.0f ? p1 : 0.0f;
            vh[lane] = v0 > 0.0f ? v0 : 0.0f;
        }
        __syncthreads();
        RAZ_NET_TICK(4);
        // policy dense 128 -> 64, lane = output: all 128 weight loads are issued before the chain
        float logit = pfc_b[lane];
#pragma unroll 1
        for (int j0 = 0; j0 < 128; j0 += 32) {  // 32 loads in flight, then their 32 chained fmas
            float wv[32];
#pragma unroll
            for (int j = 0; j < 32; ++j) wv[j] = pfc_w[(j0 + j) * 64 + lane];
#pragma unroll
            for (int j = 0; j < 32; ++j) logit = fmaf(ph[j0 + j], wv[j], logit);
        }
        RAZ_NET_TICK(5);
        // softmax: max (order-free) and the xor-butterfly sum (1,2,4,8 inside a row by DPP; the 16/32
        // steps are (r0+r1)+(r2+r3) of the four row sums)
        float m = logit;
        m = fmaxf(m, dppf<0xB1>(m));
        m = fmaxf(m, dppf<0x4E>(m));
        m = fmaxf(m, dppf<0x141>(m));
        m = fmaxf(m, dppf<0x140>(m));
        m = fmaxf(fmaxf(lanef(m, 0), lanef(m, 16)), fmaxf(lanef(m, 32), lanef(m, 48)));
        const float e = raz_det_expf(logit - m);
        float sum = e;
        sum = sum + dppf<0xB1>(sum);
        sum = sum + dppf<0x4E>(sum);
        sum = sum + dppf<0x141>(sum);
        sum = sum + dppf<0x140>(sum);
        sum = (lanef(sum, 0) + lanef(sum, 16)) + (lanef(sum, 32) + lanef(sum, 48));
        policy[(size_t)pos * 64 + lane] = e / sum;
        RAZ_NET_TICK(6);
        for (int o0 = 0; o0 < V; o0 += 64) {
            const int o = o0 + lane;
            if (o < V) {
                float acc = v1_b[o];
#pragma unroll 1
                for (int j0 = 0; j0 < 64; j0 += 32) {
                    float wv[32];
#pragma unroll
                    for (int j = 0; j < 32; ++j) wv[j] = v1_w[(j0 + j) * V + o];
#pragma unroll
                    for (int j = 0; j < 32; ++j) acc = fmaf(vh[j0 + j], wv[j], acc);
                }
                h1[o] = acc > 0.0f ? acc : 0.0f;
            }
        }
        __syncthreads();
        float acc = v2_b[0];
        for (int j = 0; j < V; ++j) acc = fmaf(h1[j], v2_w[j], acc);
        if (lane == 0) value[pos] = raz_det_tanhf(acc);
        RAZ_NET_TICK(7);
        __syncthreads();  // the next position overwrites bufT / head
    }
}

template <int F>
int launch(const float* W, int R, int V, const raz_bb* own, const raz_bb* enemy, const uint8_t* active,
           float* policy, float* value, size_t n, hipStream_t s, unsigned long long* prof) {
    const size_t shm = ((size_t)2 * F * PS + 192 + V) * sizeof(float);
    if (shm > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)k_net_mfma<F, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        if (e != hipSuccess) return raz_fail_hip(e, "raz_net_forward: hipFuncSetAttribute");
    }
    unsigned maxgrid = 2048;  // LDS-limited residency: 16 (F=16) / 8 workgroups per CU
    if (const char* g = getenv("RAZ_NET_MAXGRID"))   // tests only: fewer workgroups, so that each one loops over several positions
        if (atoi(g) > 0) maxgrid = (unsigned)atoi(g);
    const unsigned grid = (unsigned)(n < maxgrid ? n : maxgrid);
    if (prof)
        hipLaunchKernelGGL((k_net_mfma<F, true>), dim3(grid), dim3(64), shm, s, W, R, V, own, enemy, active, policy,
                           value, (int)n, prof);
    else
        hipLaunchKernelGGL((k_net_mfma<F, false>), dim3(grid), dim3(64), shm, s, W, R, V, own, enemy, active, policy,
                           value, (int)n, prof);
    return raz_check_launch("raz_net_forward (mfma)");
}

}  // namespace

bool raz_net_mfma_supported(int F, int V) { return (F == 16 || F == 32 || F == 64) && V <= 1024; }

int raz_net_forward_mfma(const float* W, int F, int R, int V, const uint64_t* own, const uint64_t* enemy,
                         const uint8_t* active, float* policy, float* value, size_t n, hipStream_t s,
                         unsigned long long* prof) {
    const raz_bb* o = (const raz_bb*)own;
    const raz_bb* e = (const raz_bb*)enemy;
    switch (F) {
        case 16: return launch<16>(W, R, V, o, e, active, policy, value, n, s, prof);
        case 32: return launch<32>(W, R, V, o, e, active, policy, value, n, s, nullptr);
        case 64: return launch<64>(W, R, V, o, e, active, policy, value, n, s, nullptr);
        default: return raz_fail(RAZ_EINVAL, "raz_net_forward_mfma: unsupported filter count");
    }
}
