// raz_net_wide.hip — forward pass of WIDE policy/value nets (F >= 128, e.g. the 256x10 net of
// config.py:187-193 used by ch5.yml / alpha_go_zero.yml) for a batch of leaf positions: every 3x3
// convolution is one launch of an implicit-GEMM kernel on v_mfma_f32_32x32x2_f32
// (D[32 ch x 32 sq] += A[32 x 2] * B[2 x 32], f32 in / f32 accumulate = a k-ordered fmaf chain, so
// results are bit-identical to k_net_wave and the CPU oracle under raznet-forward-v1).
//
// GEMM view per layer: M = out channels (F), N = squares of all positions (64 n), K = 9 F.
//   workgroup  = 4 waves = 4 positions x 64 output channels; wave = 1 position x 64 channels
//                = 2 x 2 MFMA tiles (64 accumulator VGPRs)
//   K loop     = 16-input-channel chunks (the order raznet-forward-v1 fixes): per chunk the block
//                stages (a) the 4 positions' 16 activation planes as zero-haloed LDS planes
//                (row stride 12, plane stride 136: taps are immediate offsets, no predicates) and
//                (b) the chunk's weights for its 64 output channels, pre-arranged in HBM in
//                MFMA-operand order so the copy is linear; then 72 k-steps x 4 MFMAs per wave
//   traffic    = activations [n][F][64] f32 in HBM: read F/64 times, written once per layer
//                (~1 GB per layer at n = 8192 against 0.62 TFLOP: firmly MFMA-bound);
//                weights (9 F^2 floats per layer) are re-read from L2 by every workgroup
//   epilogue   = (+ skip) , relu, coalesced 128-byte row segments (lane = square)
// Layer 0 (2 input planes straight from the bitboards) and the heads are small VALU kernels.
#include <hip/hip_runtime.h>
#include "raz_bitboard.h"
#include "raz_detmath.h"
#include "raz_internal.h"
#include "raz_net_layout.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int PS = 136;   // haloed plane stride (floats)
constexpr int KS = 72;    // k-steps (of 2) per 16-channel chunk: 9 taps x 8

__device__ __forceinline__ int pidx(int sq) { return ((sq >> 3) + 1) * 12 + (sq & 7) + 4; }

// in/out/skip: [n][F][64] f32.  grid = (ceil(n/4), F/64), block = 256.
__global__ __launch_bounds__(256) void k_conv3x3_wide(const float* __restrict__ Wl /* region 3, this layer */,
                                                      const float* __restrict__ bias, const float* in,
                                                      float* out, const float* skip, const uint8_t* __restrict__ active,
                                                      int n, int F) {
    alignas(16) static float smem[RAZ_EMU_LDS_FLOATS];
    float* actP = smem;                 // [4][16][PS]
    float* wA = smem + 4 * 16 * PS;     // [KS][2][64]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int p0 = blockIdx.x * 4, nt = blockIdx.y, ntiles = gridDim.y;
    const int pos = p0 + wv;
    const bool live = pos < n && (!active || active[pos]);
    // zero the activation planes once: halos stay zero, interiors are overwritten every chunk
    for (int j = tid; j < 4 * 16 * PS / 4; j += 256) ((f32x4*)actP)[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x16 acc[2][2];
    {
        // C/D layout of 32x32: row = (r&3) + 8*(r>>2) + 4*(lane>>5), col = lane&31
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float b = bias[nt * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)];
                acc[mt][0][r] = b;
                acc[mt][1][r] = b;
            }
    }
    const int nchunks = F / 16;
    const float* bbase = actP + (wv * 16 + (lane >> 5)) * PS + pidx(lane & 31);  // + 48 for squares 32..63
    for (int c = 0; c < nchunks; ++c) {
        __syncthreads();  // previous chunk's MFMA reads are done
        // (a) activations: 4 positions x 16 planes x 64 floats = 1024 float4, 4 per thread
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = tid + 256 * i;
            const int pp = q >> 8, ch = (q >> 4) & 15, r4 = q & 15;
            f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (p0 + pp < n) v = *(const f32x4*)(in + ((size_t)(p0 + pp) * F + c * 16 + ch) * 64 + r4 * 4);
            *(f32x4*)(actP + (pp * 16 + ch) * PS + ((r4 >> 1) + 1) * 12 + 4 + (r4 & 1) * 4) = v;
        }
        // (b) weights: 72 x 128 floats = 2304 float4, 9 per thread, linear
        const f32x4* wsrc = (const f32x4*)(Wl + ((size_t)c * ntiles + nt) * (KS * 128));
#pragma unroll
        for (int i = 0; i < 9; ++i) ((f32x4*)wA)[tid + 256 * i] = wsrc[tid + 256 * i];
        __syncthreads();
        // (c) 72 k-steps x (2 channel tiles x 2 square tiles)
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int toff = (t / 3 - 1) * 12 + (t % 3 - 1);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int s = t * 8 + j;
                const float a0 = wA[(s * 2 + 0) * 64 + lane], a1 = wA[(s * 2 + 1) * 64 + lane];
                const float b0 = bbase[2 * j * PS + toff], b1 = bbase[2 * j * PS + toff + 48];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
        }
    }
    if (!live) return;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int oc = nt * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const size_t o = ((size_t)pos * F + oc) * 64 + st * 32 + (lane & 31);
                float v = acc[mt][st][r];
                if (skip) v = v + skip[o];
                out[o] = v > 0.0f ? v : 0.0f;
            }
}

// Layer 0: 2 bit-planes -> F channels.  One wave per position, lane = square.  Weights in the
// "wave" layout [F/16][9][2][16] + bias (region 1).
__global__ __launch_bounds__(64) void k_conv0_wide(const float* __restrict__ W0, const raz_bb* __restrict__ own,
                                                   const raz_bb* __restrict__ enemy,
                                                   const uint8_t* __restrict__ active, float* out, int n, int F) {
    const int pos = blockIdx.x, lane = threadIdx.x;
    if (pos >= n || (active && !active[pos])) return;
    const raz_bb bo = own[pos], be = enemy[pos];
    const int y = lane >> 3, x = lane & 7;
    float x0[9], x1[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
        const bool ok = (yy >= 0) && (yy < 8) && (xx >= 0) && (xx < 8);
        const int s = (yy * 8 + xx) & 63;
        x0[t] = ok ? (float)((bo >> s) & 1) : 0.0f;
        x1[t] = ok ? (float)((be >> s) & 1) : 0.0f;
    }
    const float* bias = W0 + (size_t)F * 18;
    for (int ocb = 0; ocb < F / 16; ++ocb) {
        float acc[16];
#pragma unroll
        for (int o = 0; o < 16; ++o) acc[o] = bias[ocb * 16 + o];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float* wt = W0 + ((size_t)ocb * 9 + t) * 32;
#pragma unroll
            for (int o = 0; o < 16; ++o) acc[o] = fmaf(x0[t], wt[o], acc[o]);
#pragma unroll
            for (int o = 0; o < 16; ++o) acc[o] = fmaf(x1[t], wt[16 + o], acc[o]);
        }
#pragma unroll
        for (int o = 0; o < 16; ++o) out[((size_t)pos * F + ocb * 16 + o) * 64 + lane] = acc[o] > 0.0f ? acc[o] : 0.0f;
    }
}

// Heads: 1x1 convs, dense layers, softmax, tanh.  One wave per position over the trunk output in HBM.
__global__ __launch_bounds__(64) void k_heads_wide(const float* __restrict__ H, const float* trunk,
                                                   const uint8_t* __restrict__ active, float* __restrict__ policy,
                                                   float* __restrict__ value, int n, int F, int V) {
    alignas(16) static float head[RAZ_EMU_LDS_FLOATS];  // ph[128] vh[64] h1[V]
    const int pos = blockIdx.x, lane = threadIdx.x;
    if (pos >= n || (active && !active[pos])) return;
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
    const float* a = trunk + (size_t)pos * F * 64 + lane;
    float p0 = pol_b[0], p1 = pol_b[1], v0 = val_b[0];
#pragma unroll 8
    for (int ic = 0; ic < F; ++ic) {
        const float xv = a[(size_t)ic * 64];
        p0 = fmaf(xv, pol_w[ic], p0);
        p1 = fmaf(xv, pol_w[F + ic], p1);
        v0 = fmaf(xv, val_w[ic], v0);
    }
    ph[lane] = p0 > 0.0f ? p0 : 0.0f;
    ph[64 + lane] = p1 > 0.0f ? p1 : 0.0f;
    vh[lane] = v0 > 0.0f ? v0 : 0.0f;
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

}  // namespace

size_t raz_net_wide_scratch_bytes(int F, size_t n) { return (size_t)2 * n * F * 64 * sizeof(float); }

int raz_net_forward_wide(const float* W, int F, int R, int V, const uint64_t* own, const uint64_t* enemy,
                         const uint8_t* active, float* policy, float* value, size_t n, void* scratch,
                         size_t scratch_bytes, hipStream_t s) {
    if (!scratch || scratch_bytes < raz_net_wide_scratch_bytes(F, n))
        return raz_fail(RAZ_ENOMEM, "raz_net_forward: scratch too small (raz_net_scratch_bytes)");
    float* bufA = (float*)scratch;
    float* bufT = bufA + (size_t)n * F * 64;
    const size_t shm = ((size_t)4 * 16 * PS + KS * 128) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_conv3x3_wide, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        if (e != hipSuccess) return raz_fail_hip(e, "raz_net_forward: hipFuncSetAttribute");
        attr_set = true;
    }
    hipLaunchKernelGGL(k_conv0_wide, dim3((unsigned)n), dim3(64), 0, s, W + conv_off(F, 0), (const raz_bb*)own,
                       (const raz_bb*)enemy, active, bufA, (int)n, F);
    const dim3 grid((unsigned)((n + 3) / 4), (unsigned)(F / 64));
    for (int r = 0; r < R; ++r) {
        const int l1 = 1 + 2 * r, l2 = 2 + 2 * r;
        hipLaunchKernelGGL(k_conv3x3_wide, grid, dim3(256), shm, s, W + wide_tile_off(F, R, V, l1, 0, 0),
                           W + conv_off(F, l1) + (size_t)F * 9 * F, (const float*)bufA, bufT, (const float*)nullptr,
                           active, (int)n, F);
        hipLaunchKernelGGL(k_conv3x3_wide, grid, dim3(256), shm, s, W + wide_tile_off(F, R, V, l2, 0, 0),
                           W + conv_off(F, l2) + (size_t)F * 9 * F, (const float*)bufT, bufA, (const float*)bufA,
                           active, (int)n, F);
    }
    hipLaunchKernelGGL(k_heads_wide, dim3((unsigned)n), dim3(64), (192 + (size_t)V) * sizeof(float), s,
                       W + heads_off(F, R), (const float*)bufA, active, policy, value, (int)n, F, V);
    return raz_check_launch("raz_net_forward (wide)");
}
