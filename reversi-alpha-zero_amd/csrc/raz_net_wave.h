// raz_net_wave.h — device code shared by the narrow-net kernels (raz_net_mfma.hip) and the fused tree + net kernel of the
// engine (raz_engine.hip k_tree_net): the zero-haloed LDS plane layout, one 3x3 conv layer on v_mfma_f32_16x16x4_f32, and the
// whole forward pass of ONE position of an F == 16 net by ONE wave (raz_net16_forward_in_wave).  Everything here computes
// raznet-forward-v1: every output one k-ordered fmaf chain, bit-identical to the CPU oracle.
#pragma once
#include <hip/hip_runtime.h>
#include "raz_bitboard.h"
#include "raz_detmath.h"
#include "raz_net_layout.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int PS = 136;  // padded plane stride in floats (136 % 32 == 8 keeps bank overlap at 4 of 32)

__device__ __forceinline__ int pidx(int sq) { return ((sq >> 3) + 1) * 12 + (sq & 7) + 4; }

template <int CTRL>
__device__ __forceinline__ float dppf(float v) {
    return __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float lanef(float v, int l) {
    return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), l));
}

template <int F, int CIN, bool FIRST>
struct LayerK {
    static constexpr int KS = FIRST ? 5 : 9 * CIN / 4;
};

template <int KS>
__device__ __forceinline__ void load_wregs(const float* __restrict__ Wl, int nt, int lane, float (&wreg)[KS]) {
#pragma unroll
    for (int s = 0; s < KS; ++s) wreg[s] = Wl[((size_t)nt * KS + s) * 64 + lane];
}

// LDS hand-over inside ONE wave (single-wave workgroups)
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One 3x3 conv layer over the zero-haloed planes `in` (CIN channels; for FIRST the two input bit
// planes) -> `out` (F channels).  SKIP: out is also the residual input (updated in place).
// PRE: the layer's B operands are already in registers (`pre`, F == 16 only).
// INPLACE (F == 16, one channel tile: raz_net16_forward_in_wave): `in` and `out` are the SAME buffer - a single wave performs every
// operand read of the layer before its epilogue stores; the residual input is the lane's own D fragment of the previous block
// (`frag`, kept in registers; keep_frag: this layer's output becomes the next residual), and the closing synchronisation is
// wave-local.
// MT: the 16-square M tiles the wave computes, starting at `in` / `out` (4 = the whole board; the product uses nothing else).
template <int F, int CIN, bool FIRST, bool SKIP, bool PRE, int MT = 4, bool INPLACE = false>
__device__ __forceinline__ void conv_layer(const float* __restrict__ Wl, const float* __restrict__ bias,
                                           const float* in, float* out, int lane,
                                           const float (&pre)[LayerK<F, CIN, FIRST>::KS], float preb,
                                           f32x4* frag = nullptr, bool keep_frag = false) {
    constexpr int KS = LayerK<F, CIN, FIRST>::KS;
    const int i = lane & 15, kk = lane >> 4;
    for (int nt = 0; nt < F / 16; ++nt) {
        float wreg[KS];
        float b;
        if (PRE) {
#pragma unroll
            for (int s = 0; s < KS; ++s) wreg[s] = pre[s];
            b = preb;
        } else {
            load_wregs<KS>(Wl, nt, lane, wreg);
            b = bias[nt * 16 + i];
        }
        f32x4 acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = (f32x4){b, b, b, b};
        if (FIRST) {
            // k = tap*2 + plane: k-step s holds taps 2s (lanes 0..31) and 2s+1 (lanes 32..63); the
            // padded k = 18, 19 carry zero weights, so any in-bounds address will do for them
            const float* base = in + (kk & 1) * PS + pidx(i);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int ta = 2 * s, tb = 2 * s + 1 > 8 ? 8 : 2 * s + 1;
                const int offa = (ta / 3 - 1) * 12 + (ta % 3 - 1), offb = (tb / 3 - 1) * 12 + (tb % 3 - 1);
                const int off = (kk < 2) ? offa : offb;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const float a = base[off + mt * 24];
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wreg[s], acc[mt], 0, 0, 0);
                }
            }
        } else {
            const float* base = in + kk * PS + pidx(i);  // M tile mt adds 24 floats (two board rows)
            // k order of raznet-forward-v1: 16-channel chunks, then tap, then channel within the chunk.
            // Software-pipelined DEPTH k-steps deep: the A operands of k-step s + DEPTH are requested before the MFMAs of k-step s, so
            // an LDS round trip has DEPTH x 4 x 32 matrix-core cycles to land (same MFMAs in the same order: bit-identical).
            auto koff = [](int s) {
                const int c = s / 36, t = (s / 4) % 9, q = s % 4;
                return (c * 16 + q * 4) * PS + (t / 3 - 1) * 12 + (t % 3 - 1);
            };
            constexpr int DEPTH = 2;   // k-steps of A operands in flight ahead of the one being multiplied
            float aq[DEPTH + 1][MT];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) aq[d][mt] = base[koff(d < KS ? d : KS - 1) + mt * 24];
            }
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if (s + DEPTH < KS) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) aq[DEPTH][mt] = base[koff(s + DEPTH) + mt * 24];
                }
                // keep the requests above where they are: the matrix-core instructions below must not be hoisted over them
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[0][mt], wreg[s], acc[mt], 0, 0, 0);
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) aq[d][mt] = aq[d + 1][mt];
                }
            }
        }
        // D fragment: channel nt*16 + i, squares mt*16 + kk*4 + r (r = 0..3): one row segment
        float* obase = out + (nt * 16 + i) * PS + ((kk >> 1) + 1) * 12 + (kk & 1) * 4 + 4;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            f32x4* dst = (f32x4*)(obase + mt * 24);
            f32x4 v = acc[mt];
            if (SKIP) v = v + (INPLACE ? frag[mt] : *dst);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.0f ? v[r] : 0.0f;
            *dst = v;
            if (INPLACE && keep_frag) frag[mt] = v;
        }
    }
    if (INPLACE)
        wave_lds_sync();
    else
        __syncthreads();
}


// The forward pass of ONE position of an F == 16 net by the calling wave, for a kernel that already runs one wave per game
// (k_tree_net).  `buf`: 16 * PS + 192 + V floats of LDS owned by this wave whose first 16 * PS floats were zeroed once
// (raz_net16_lds_floats / raz_net16_zero_planes; the halos must read as 0, the interiors are overwritten for every position).
// The trunk runs in place in that single plane buffer (conv_layer<.., INPLACE>), the layers' B operands are fetched from L2
// when the layer starts (no registers stay resident between positions: the caller's own state has to fit beside this in 128
// VGPRs for 4 waves per SIMD).  Returns the policy entry of square `lane` and the value.  Same operations in the same order as
// k_net_mfma: bit-identical.
__host__ __device__ inline constexpr int raz_net16_lds_floats(int V) { return 16 * PS + 192 + V; }
__device__ __forceinline__ void raz_net16_zero_planes(float* buf, int lane) {
    f32x4* z = (f32x4*)buf;
    for (int j = lane; j < 16 * PS / 4; j += 64) z[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    wave_lds_sync();
}
__device__ __forceinline__ void raz_net16_forward_in_wave(const float* __restrict__ W, int R, int V, raz_bb bo, raz_bb be, float* buf,
                                                          int lane, float& policy_of_lane, float& value) {
    constexpr int F = 16;
    float* head = buf + F * PS;  // ph[128] vh[64] h1[V]
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
    buf[pidx(lane)] = (float)((bo >> lane) & 1ULL);       // the two input bit planes: planes 0/1
    buf[PS + pidx(lane)] = (float)((be >> lane) & 1ULL);
    wave_lds_sync();
    f32x4 frag[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) frag[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    {
        const float dummy5[5] = {0, 0, 0, 0, 0};
        float dummyK[LayerK<F, F, false>::KS];
        conv_layer<F, 2, true, false, false, 4, true>(W + mfma_layer_off(F, R, V, 0), W + conv_off(F, 0) + (size_t)F * 9 * 2, buf, buf, lane,
                                                      dummy5, 0.f, frag, true);
        for (int r = 0; r < R; ++r) {
            const int l1 = 1 + 2 * r, l2 = 2 + 2 * r;
            conv_layer<F, F, false, false, false, 4, true>(W + mfma_layer_off(F, R, V, l1), W + conv_off(F, l1) + (size_t)F * 9 * F, buf, buf,
                                                           lane, dummyK, 0.f, frag, false);
            conv_layer<F, F, false, true, false, 4, true>(W + mfma_layer_off(F, R, V, l2), W + conv_off(F, l2) + (size_t)F * 9 * F, buf, buf,
                                                          lane, dummyK, 0.f, frag, true);
        }
    }
    {
        const float* a = buf + pidx(lane);
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
    wave_lds_sync();
    float logit = pfc_b[lane];
#pragma unroll 1
    for (int j0 = 0; j0 < 128; j0 += 32) {  // 32 loads in flight, then their 32 chained fmas
        float wv[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) wv[j] = pfc_w[(j0 + j) * 64 + lane];
#pragma unroll
        for (int j = 0; j < 32; ++j) logit = fmaf(ph[j0 + j], wv[j], logit);
    }
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
    policy_of_lane = e / sum;
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
    wave_lds_sync();
    float acc = v2_b[0];
    for (int j = 0; j < V; ++j) acc = fmaf(h1[j], v2_w[j], acc);
    value = raz_det_tanhf(acc);
    wave_lds_sync();  // the next position overwrites the planes / head
}

}  // namespace
