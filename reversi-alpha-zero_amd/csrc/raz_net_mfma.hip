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

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int PS = 136;  // padded plane stride in floats (136 % 32 == 8 keeps bank overlap at 4 of 32)

__device__ __forceinline__ int pidx(int sq) { return ((sq >> 3) + 1) * 12 + (sq & 7) + 4; }

// One 3x3 conv layer over the zero-haloed planes `in` (CIN channels) -> `out` (F channels).
// SKIP: out is also the residual input (updated in place).  FIRST: input planes from bitboards.
template <int F, int CIN, bool FIRST, bool SKIP>
__device__ __forceinline__ void conv_layer(const float* __restrict__ Wl, const float* __restrict__ bias,
                                           const float* in, float* out, raz_bb bo, raz_bb be, int lane) {
    constexpr int KS = FIRST ? 5 : 9 * CIN / 4;
    const int i = lane & 15, kk = lane >> 4;
    for (int nt = 0; nt < F / 16; ++nt) {
        float wreg[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) wreg[s] = Wl[((size_t)nt * KS + s) * 64 + lane];
        const float b = bias[nt * 16 + i];
        f32x4 acc[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt] = (f32x4){b, b, b, b};
        if (FIRST) {
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int k = 4 * s + kk, t = k >> 1;
                const raz_bb board = (k & 1) ? be : bo;
                const int dy = t / 3 - 1, dx = t % 3 - 1;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const int y = mt * 2 + (i >> 3) + dy, x = (i & 7) + dx;
                    const bool ok = (k < 18) && (y >= 0) && (y < 8) && (x >= 0) && (x < 8);
                    const float a = ok ? (float)((board >> ((y * 8 + x) & 63)) & 1ULL) : 0.0f;
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wreg[s], acc[mt], 0, 0, 0);
                }
            }
        } else {
            const float* base = in + kk * PS + pidx(i);  // M tile mt adds 24 floats (two board rows)
            // k order of raznet-forward-v1: 16-channel chunks, then tap, then channel within the chunk
#pragma unroll
            for (int c = 0; c < CIN / 16; ++c) {
#pragma unroll
                for (int t = 0; t < 9; ++t) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int s = (c * 9 + t) * 4 + q;
                        const int off = (c * 16 + q * 4) * PS + (t / 3 - 1) * 12 + (t % 3 - 1);
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt) {
                            const float a = base[off + mt * 24];
                            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wreg[s], acc[mt], 0, 0, 0);
                        }
                    }
                }
            }
        }
        // D fragment: channel nt*16 + i, squares mt*16 + kk*4 + r (r = 0..3): one row segment
        float* obase = out + (nt * 16 + i) * PS + ((kk >> 1) + 1) * 12 + (kk & 1) * 4 + 4;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            f32x4* dst = (f32x4*)(obase + mt * 24);
            f32x4 v = acc[mt];
            if (SKIP) {
                const f32x4 sk = *dst;
                v = v + sk;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.0f ? v[r] : 0.0f;
            *dst = v;
        }
    }
    __syncthreads();
}

template <int F>
__global__ __launch_bounds__(64) void k_net_mfma(const float* __restrict__ W, int R, int V,
                                                 const raz_bb* __restrict__ own,
                                                 const raz_bb* __restrict__ enemy,
                                                 const uint8_t* __restrict__ active,
                                                 float* __restrict__ policy, float* __restrict__ value, int n) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int pos = blockIdx.x;
    if (pos >= n) return;
    if (active && !active[pos]) return;
    const int lane = threadIdx.x;
    float* bufA = smem;
    float* bufT = smem + F * PS;
    float* head = smem + 2 * F * PS;  // ph[128] vh[64] h1[V]
    {  // zero both buffers: the halo must read as 0
        f32x4* z = (f32x4*)smem;
        for (int j = lane; j < 2 * F * PS / 4; j += 64) z[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    const raz_bb bo = own[pos], be = enemy[pos];
    const float* Wm = W;  // mfma region offsets are absolute (raz_net_layout.h)
    conv_layer<F, 2, true, false>(Wm + mfma_layer_off(F, R, V, 0), W + conv_off(F, 0) + (size_t)F * 9 * 2, nullptr,
                                  bufA, bo, be, lane);
    for (int r = 0; r < R; ++r) {
        const int l1 = 1 + 2 * r, l2 = 2 + 2 * r;
        conv_layer<F, F, false, false>(Wm + mfma_layer_off(F, R, V, l1), W + conv_off(F, l1) + (size_t)F * 9 * F, bufA,
                                       bufT, 0, 0, lane);
        conv_layer<F, F, false, true>(Wm + mfma_layer_off(F, R, V, l2), W + conv_off(F, l2) + (size_t)F * 9 * F, bufT,
                                      bufA, 0, 0, lane);
    }
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
    float logit = pfc_b[lane];
#pragma unroll 16
    for (int j = 0; j < 128; ++j) logit = fmaf(ph[j], pfc_w[j * 64 + lane], logit);
    float m = logit;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) m = fmaxf(m, __shfl_xor(m, s));
    const float e = raz_det_expf(logit - m);
    float sum = e;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) sum = sum + __shfl_xor(sum, s);
    policy[(size_t)pos * 64 + lane] = e / sum;
    for (int o0 = 0; o0 < V; o0 += 64) {
        const int o = o0 + lane;
        if (o < V) {
            float acc = v1_b[o];
#pragma unroll 8
            for (int j = 0; j < 64; ++j) acc = fmaf(vh[j], v1_w[j * V + o], acc);
            h1[o] = acc > 0.0f ? acc : 0.0f;
        }
    }
    __syncthreads();
    float acc = v2_b[0];
    for (int j = 0; j < V; ++j) acc = fmaf(h1[j], v2_w[j], acc);
    if (lane == 0) value[pos] = raz_det_tanhf(acc);
}

template <int F>
int launch(const float* W, int R, int V, const raz_bb* own, const raz_bb* enemy, const uint8_t* active,
           float* policy, float* value, size_t n, hipStream_t s) {
    const size_t shm = ((size_t)2 * F * PS + 192 + V) * sizeof(float);
    if (shm > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)k_net_mfma<F>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        if (e != hipSuccess) return raz_fail_hip(e, "raz_net_forward: hipFuncSetAttribute");
    }
    hipLaunchKernelGGL(k_net_mfma<F>, dim3((unsigned)n), dim3(64), shm, s, W, R, V, own, enemy, active, policy, value,
                       (int)n);
    return raz_check_launch("raz_net_forward (mfma)");
}

}  // namespace

bool raz_net_mfma_supported(int F, int V) { return (F == 16 || F == 32 || F == 64) && V <= 1024; }

int raz_net_forward_mfma(const float* W, int F, int R, int V, const uint64_t* own, const uint64_t* enemy,
                         const uint8_t* active, float* policy, float* value, size_t n, hipStream_t s) {
    const raz_bb* o = (const raz_bb*)own;
    const raz_bb* e = (const raz_bb*)enemy;
    switch (F) {
        case 16: return launch<16>(W, R, V, o, e, active, policy, value, n, s);
        case 32: return launch<32>(W, R, V, o, e, active, policy, value, n, s);
        case 64: return launch<64>(W, R, V, o, e, active, policy, value, n, s);
        default: return raz_fail(RAZ_EINVAL, "raz_net_forward_mfma: unsupported filter count");
    }
}
