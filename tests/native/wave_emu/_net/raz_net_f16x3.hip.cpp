// raz_net_f16x3.hip — forward pass of WIDE policy/value nets (F % 128 == 0, e.g. the 256x10 net of config.py:187-193)
// with the 3x3 convolutions of the trunk on the f16 matrix cores at f32-class accuracy: "raznet-forward-v2".
//
// f32 MFMA runs at the f32 vector rate (157 TF); f16 MFMA at 16x that.  Every f32 operand x is carried as a pair of halfs
// (hi, lo) = (f16(x), f16(x - hi)) - 22 significant bits; subnormal halfs are kept by the matrix core (probed:
// tools/probe_f16.hip) - and a product x*w is evaluated as  hi_x*hi_w + hi_x*lo_w + lo_x*hi_w  (the dropped lo*lo term is
// 2^-22 relative), three v_mfma_f32_32x32x16_f16 accumulating in f32 into ONE accumulator.  Weights are pre-scaled per
// layer by a power of two S (max |w| * S in [2^14, 2^15): the lo halves of all but negligible weights stay normal numbers;
// the accumulator is multiplied by the exact 1/S before the bias).  Net effect: 3/16 of the f32-MFMA time per MAC, results
// within 1e-5 of the fp32 graph (tests: vs fp32 torch and vs the exact-f32 kernel raznet-forward-v1 on the benchmarked
// shape), NOT bit-identical to the CPU oracle's fmaf chains - the matrix core's internal 16-term summation is not a
// documented IEEE sequence - so games played on this path are checked against the oracle fed with THIS net's outputs
// through the reference's own NN seam (ReversiPlayer(api=...), agent/player.py:41).  The first layer (2 planes) and the
// heads stay exact-f32 VALU chains as in v1.
//
// Activations live in HBM already split, in the order the kernel's LDS image wants them:
//     [position][16-channel chunk][plane: k-group(8 ch) x {hi, lo} = 4][square 64][8 halfs]      (F*256 bytes per position)
// so that one (position, chunk) is four lane-linear 1 KiB pieces moved by global_load_lds (no registers, no ds_write).
//
// GEMM view per layer: D[oc, sq] += W[oc, k] * X[k, sq], k = (chunk, tap, channel in chunk).
//   workgroup = 8 waves = 128 output channels x 8 positions (one workgroup per CU, 2 waves per SIMD, <= 256 VGPRs);
//               wave = 128 oc x 64 squares of ONE position = 4 x 2 MFMA tiles (128 accumulator registers)
//   K loop    = 16 chunks x 3 tap groups = 48 stages; a stage's weights (24 KB: 3 taps x 16 channels x 128 oc x {hi, lo})
//               and, once per chunk, the 8 positions' activations (32 KB) arrive by LDS-DMA in a double buffer while the
//               previous stage's 3 taps x 8 tiles x 3 MFMAs per wave run: one barrier per stage, no register staging
//   taps      = per-lane LDS addresses precomputed once (18 VGPRs): square + tap shift, or - off the board - a slot of a
//               zero row in the same bank class, so there are no predicates and no halo in the image
//   LDS reads = ds_read_b128, conflict-free by construction: lanes 0-31 read consecutive 16-byte slots (squares or channels),
//               lanes 32-63 the other k-group's plane; 0.5 reads per MFMA
//   epilogue  = * 1/S, + bias, (+ skip), relu, split into (hi, lo), 8-byte stores that tile 512-byte runs
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "raz_bitboard.h"
#include "raz_detmath.h"
#include "raz_internal.h"
#include "raz_net_layout.h"

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int OCT = 128;                         // output channels per workgroup
constexpr int NWAVE = 8;                         // waves per workgroup = positions per workgroup
constexpr int W_STAGE = 3 * 2 * 2 * OCT * 16;    // 24,576 B: [tap 3][k-group 2][hi/lo][oc 128][16 B]
constexpr int ACT_POS = 4 * 64 * 16;             // 4,096 B: [plane 4][square 64][16 B]
constexpr int ACT_IMG = NWAVE * ACT_POS + 1280;  // one chunk's image: 8 positions + two zero rows of 256 B, 1,024 B apart
constexpr int LDS_W = 0;                         // two weight stages (double buffer)
constexpr int LDS_ACT = 2 * W_STAGE;             // two activation images (double buffer)
constexpr int LDS_BYTES = LDS_ACT + 2 * ACT_IMG; // 117,248 B: one workgroup of 8 waves per CU
constexpr int Z_OFF = NWAVE * ACT_POS;           // the zero rows inside an image (at Z_OFF and Z_OFF + 1024)

// Timeline instrumentation of the conv kernel, compiled in ONLY by tools/probe_conv.hip (which includes this file with
// RAZ_F16X3_STAMPS defined): per wave 8 words - s_memtime at entry [0], when stage 0 has landed [1], at the end of the K loop [2],
// after the last store was issued [3] and after the stores have drained [4]; the cycles spent inside the 48 stage barriers [5];
// HW_ID [6] (which CU / SIMD ran the wave); s_memrealtime at entry [7].  The product build has none of it.
#ifdef RAZ_F16X3_STAMPS
#define RAZ_STAMP_PARAM , unsigned long long* __restrict__ stamps
#define RAZ_STAMP_ARG , (unsigned long long*)nullptr
#define RAZ_STAMP_BEGIN                                                                                           \
    unsigned long long st_t[8];                                                                                   \
    st_t[0] = __builtin_amdgcn_s_memtime();                                                                       \
    st_t[7] = __builtin_amdgcn_s_memrealtime();                                                                   \
    st_t[6] = (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32); \
    st_t[5] = 0;                                                                                                  \
    st_t[1] = st_t[2] = st_t[3] = st_t[4] = 0;                                                                    \
    unsigned long long st_b = 0
#define RAZ_STAMP_BARRIER_IN st_b = __builtin_amdgcn_s_memtime()
#define RAZ_STAMP_BARRIER_OUT(st_)                                    \
    do {                                                              \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();   \
        st_t[5] += t_ - st_b;                                         \
        if ((st_) == 0) st_t[1] = t_;                                 \
    } while (0)
#define RAZ_STAMP_AT(i_) st_t[i_] = __builtin_amdgcn_s_memtime()
#define RAZ_STAMP_END                                                                                    \
    do {                                                                                                 \
        st_t[3] = __builtin_amdgcn_s_memtime();                                                          \
        __builtin_amdgcn_s_waitcnt(0x0070); /* vmcnt(0) */                                               \
        st_t[4] = __builtin_amdgcn_s_memtime();                                                          \
        if (lane == 0 && stamps)                                                                         \
            for (int i_ = 0; i_ < 8; ++i_) stamps[((size_t)blockIdx.x * NWAVE + wv) * 8 + i_] = st_t[i_]; \
    } while (0)
#else
#define RAZ_STAMP_PARAM
#define RAZ_STAMP_ARG
#define RAZ_STAMP_BEGIN
#define RAZ_STAMP_BARRIER_IN
#define RAZ_STAMP_BARRIER_OUT(st_)
#define RAZ_STAMP_AT(i_)
#define RAZ_STAMP_END
#endif

#define GLDS16(gptr, lptr)                                                                                      \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),                     \
                                     (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

// in / out / skip: split activations (header).  Wl: this layer's region-4 weights.  grid = ceil(n / 8) * F / 128, block 512.
// Pipeline: stage s+1 (the next tap group's weights, and with the first tap group of a chunk that chunk's activations) is
// in flight as LDS-DMA into the other buffer while stage s is computed; ONE barrier per stage (it also drains this wave's
// share of the DMA issued a whole stage earlier).
__global__ __launch_bounds__(512, 2) void k_conv3x3_f16x3(const unsigned char* __restrict__ Wl, const float* __restrict__ bias,
                                                          const float* __restrict__ inv_scale_ptr, const unsigned char* in, unsigned char* out,
                                                          const unsigned char* skip, const uint8_t* __restrict__ active, int n, int F,
                                                          unsigned* __restrict__ flag, const uint32_t* __restrict__ n_ptr RAZ_STAMP_PARAM) {
    alignas(16) static unsigned char lds[RAZ_EMU_LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    RAZ_STAMP_BEGIN;
    if (n_ptr) n = (int)*n_ptr < n ? (int)*n_ptr : n;   // rows 0..n-1 of a compacted batch (raz_leaf_cache.hip): the count lives on the device
    const int noct = F / OCT, nchunks = F / 16;
    // blocks b and b + 8 run on the same XCD (round-robin dispatch): give them the oc tiles of the SAME positions, so the
    // later one finds the activations in that XCD's L2
    const int b = blockIdx.x;
    const int ot = (b >> 3) % noct;
    const int pg = (b / (8 * noct)) * 8 + (b & 7);
    const int p0 = pg * NWAVE, pos = p0 + wv;
    if (p0 >= n) return;
    const bool live = pos < n && (!active || active[pos]);
    const size_t pos_bytes = (size_t)F * 256;
    const unsigned char* in_pos = in + (size_t)(pos < n ? pos : n - 1) * pos_bytes;
    if (tid < 160) {   // the zero rows of both images
        const int im = tid / 80, k = tid % 80;
        ((f32x4*)(lds + LDS_ACT + im * ACT_IMG + Z_OFF))[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    // per-lane LDS byte offsets (inside an activation image) of the B operand for the 9 taps x 2 square tiles
    const int kg = lane >> 5;
    uint32_t boff[2][9];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int sq = nt * 32 + (lane & 31), y = sq >> 3, x = sq & 7;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
            const bool ok = yy >= 0 && yy < 8 && xx >= 0 && xx < 8;
            const int s2 = sq + (t / 3 - 1) * 8 + (t % 3 - 1);
            boff[nt][t] = ok ? (uint32_t)(wv * ACT_POS + kg * 2048 + s2 * 16) : (uint32_t)(Z_OFF + (s2 & 15) * 16);
        }
    }
    const uint32_t aoff = (uint32_t)(kg * 4096 + (lane & 31) * 16);   // + tap*8192 + hl*2048 + mtile*512 inside a weight stage
    f32x16 acc[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][nt][r] = 0.f;
    const unsigned char* wsrc = Wl + (size_t)ot * nchunks * 3 * W_STAGE + lane * 16;
    const unsigned char* asrc = in_pos + lane * 16;
    // stage `st` = (chunk st / 3, tap group st % 3): 24 weight pieces of 1 KiB (3 per wave) + with tap group 0 this wave's
    // position's four activation planes of that chunk
    auto issue = [&](int c, int tg) {
        const int st = c * 3 + tg;
        const unsigned char* src = wsrc + (size_t)st * W_STAGE;
        unsigned char* dst = lds + LDS_W + (st & 1) * W_STAGE;
#pragma unroll
        for (int i = 0; i < 3; ++i) GLDS16(src + (wv * 3 + i) * 1024, dst + (wv * 3 + i) * 1024);
        if (tg == 0) {
            const unsigned char* a = asrc + (size_t)c * ACT_POS;
            unsigned char* ad = lds + LDS_ACT + (c & 1) * ACT_IMG + wv * ACT_POS;
#pragma unroll
            for (int pl = 0; pl < 4; ++pl) GLDS16(a + pl * 1024, ad + pl * 1024);
        }
    };
    issue(0, 0);
    for (int c = 0; c < nchunks; ++c) {
        const uint32_t abase = (uint32_t)(LDS_ACT + (c & 1) * ACT_IMG);
#pragma unroll
        for (int tg = 0; tg < 3; ++tg) {
            const int st = c * 3 + tg;
            RAZ_STAMP_BARRIER_IN;
            __syncthreads();   // stage st has landed (every wave drained its DMA before arriving); stage st-1's reads are done
            RAZ_STAMP_BARRIER_OUT(st);
            if (tg < 2) issue(c, tg + 1);
            else if (c + 1 < nchunks) issue(c + 1, 0);
            const uint32_t wbase = (uint32_t)(LDS_W + (st & 1) * W_STAGE) + aoff;
#pragma unroll
            for (int tt = 0; tt < 3; ++tt) {
                const int t = tg * 3 + tt;
                h8 bh[2], bl[2];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    bh[nt] = *(const h8*)(lds + abase + boff[nt][t]);
                    bl[nt] = *(const h8*)(lds + abase + boff[nt][t] + 1024);
                }
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const h8 ah = *(const h8*)(lds + wbase + tt * 8192 + m * 512);
                    const h8 al = *(const h8*)(lds + wbase + tt * 8192 + m * 512 + 2048);
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        acc[m][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[nt], acc[m][nt], 0, 0, 0);
                        acc[m][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[nt], acc[m][nt], 0, 0, 0);
                        acc[m][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[nt], acc[m][nt], 0, 0, 0);
                    }
                }
            }
        }
    }
    RAZ_STAMP_AT(2);
    if (!live) return;
    const float inv_scale = *inv_scale_ptr;
    // epilogue.  D layout: column = lane & 31 = square, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) = channel in the 32-tile
    unsigned char* out_pos = out + (size_t)pos * pos_bytes;
    const unsigned char* skip_pos = skip ? skip + (size_t)pos * pos_bytes : nullptr;
    bool over = false;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int oc8 = ot * OCT + m * 32 + q * 8;   // this lane pair's 8-channel group; this lane holds 4 of them
            const f32x4 bv = *(const f32x4*)(bias + oc8 + 4 * kg);
            const size_t unit = (size_t)(oc8 >> 4) * ACT_POS + (size_t)((oc8 >> 3) & 1) * 2048;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const size_t o = unit + (size_t)(nt * 32 + (lane & 31)) * 16 + kg * 8;
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = acc[m][nt][q * 4 + j] * inv_scale + bv[j];
                if (skip_pos) {
                    const h4 sh = *(const h4*)(skip_pos + o), sl = *(const h4*)(skip_pos + o + 1024);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = v[j] + ((float)sh[j] + (float)sl[j]);
                }
                h4 hi, lo;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float r = v[j] > 0.0f ? v[j] : 0.0f;
                    over |= !(r < 60000.0f);
                    hi[j] = (_Float16)r;
                    lo[j] = (_Float16)(r - (float)hi[j]);
                }
                *(h4*)(out_pos + o) = hi;
                *(h4*)(out_pos + o + 1024) = lo;
            }
        }
    if (over) flag[(size_t)pos * RAZ_NET_ROWFLAG_WORDS] = 1u;   // an activation beyond the f16 range: this ROW is evaluated again by the exact-f32 kernel (raz_net_repair_rows)
    RAZ_STAMP_END;
}

// Layer 0: 2 bit planes -> F channels, exact f32 chains as in k_conv0_wide, written in the split layout.  The work per
// position is tiny and latency-bound (scalar weight loads), so a position's 16-channel chunks are spread over the 4 waves
// of a workgroup (lane = square, wave w takes chunks w, w + 4, ...).
__global__ __launch_bounds__(256) void k_conv0_split(const float* __restrict__ W0, const raz_bb* __restrict__ own,
                                                     const raz_bb* __restrict__ enemy, const uint8_t* __restrict__ active,
                                                     unsigned char* out, int n, int F, unsigned* __restrict__ flag,
                                                     const uint32_t* __restrict__ list, const uint32_t* __restrict__ n_ptr) {
    const int pos = blockIdx.x, lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: the weight reads below stay scalar loads
    // the forward's first kernel clears the per-row range flags (word 0 of every row's area) and the repair counter (word 1 of row 0's)
    if (threadIdx.x == 0) {
        flag[(size_t)pos * RAZ_NET_ROWFLAG_WORDS] = 0u;
        if (pos == 0) flag[1] = 0u;
    }
    __syncthreads();
    if (n_ptr) n = (int)*n_ptr < n ? (int)*n_ptr : n;
    if (pos >= n || (!list && active && !active[pos])) return;
    bool over = false;
    const size_t src = list ? list[pos] : (size_t)pos;   // compacted batch: row `pos` holds the leaf of exchange row list[pos]
    const raz_bb bo = own[src], be = enemy[src];
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
    unsigned char* op = out + (size_t)pos * F * 256;
    for (int ocb = wv; ocb < F / 16; ocb += 4) {
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
        for (int g = 0; g < 2; ++g) {
            h8 hi, lo;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float r = acc[g * 8 + j] > 0.0f ? acc[g * 8 + j] : 0.0f;
                over |= !(r < 60000.0f);
                hi[j] = (_Float16)r;
                lo[j] = (_Float16)(r - (float)hi[j]);
            }
            *(h8*)(op + (size_t)ocb * ACT_POS + (g * 2 + 0) * 1024 + lane * 16) = hi;
            *(h8*)(op + (size_t)ocb * ACT_POS + (g * 2 + 1) * 1024 + lane * 16) = lo;
        }
    }
    if (over) flag[(size_t)pos * RAZ_NET_ROWFLAG_WORDS] = 1u;
}

// Heads as in k_heads_wide (exact f32 chains), reading the trunk output in the split layout: x = hi + lo (exact in f32).
__global__ __launch_bounds__(64) void k_heads_split(const float* __restrict__ H, const unsigned char* trunk,
                                                    const uint8_t* __restrict__ active, float* __restrict__ policy,
                                                    float* __restrict__ value, int n, int F, int V,
                                                    const uint32_t* __restrict__ list, const uint32_t* __restrict__ n_ptr) {
    alignas(16) static float head[RAZ_EMU_LDS_FLOATS];  // ph[128] vh[64] h1[V]
    const int pos = blockIdx.x, lane = threadIdx.x;
    if (n_ptr) n = (int)*n_ptr < n ? (int)*n_ptr : n;
    if (pos >= n || (!list && active && !active[pos])) return;
    const size_t dst = list ? list[pos] : (size_t)pos;   // results go back to the leaf-exchange row
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
    const unsigned char* a = trunk + (size_t)pos * F * 256 + lane * 16;
    float p0 = pol_b[0], p1 = pol_b[1], v0 = val_b[0];
    for (int c = 0; c < F / 16; ++c) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const h8 hi = *(const h8*)(a + (size_t)c * ACT_POS + (g * 2 + 0) * 1024);
            const h8 lo = *(const h8*)(a + (size_t)c * ACT_POS + (g * 2 + 1) * 1024);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int ic = c * 16 + g * 8 + j;
                const float xv = (float)hi[j] + (float)lo[j];
                p0 = fmaf(xv, pol_w[ic], p0);
                p1 = fmaf(xv, pol_w[F + ic], p1);
                v0 = fmaf(xv, val_w[ic], v0);
            }
        }
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
    policy[dst * 64 + lane] = e / sum;
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
    if (lane == 0) value[dst] = raz_det_tanhf(acc);
}

}  // namespace

// Host side of raz_net_load for region 4: `src` = the blob's float parameters, `dst` = the device image being built.
void raz_net_build_f16x3(const float* src, float* dst, int F, int R, int V) {
    const float* lsrc = src + ((size_t)F * 18 + F);   // layer 1
    float* scales = dst + f16x3_scale_off(F, R, V);
    const int nchunks = F / 16, noct = F / 128;
    for (int l = 1; l < 2 * R + 1; ++l) {
        float mx = 0.f;
        for (size_t i = 0; i < (size_t)F * F * 9; ++i) mx = fmaxf(mx, fabsf(lsrc[i]));
        int e = 0;
        if (mx > 0.f) frexpf(mx, &e);                 // mx = f * 2^e, f in [0.5, 1)  =>  mx * 2^(15 - e) in [2^14, 2^15)
        const float S = ldexpf(1.0f, 15 - e);
        scales[l - 1] = ldexpf(1.0f, e - 15);
        _Float16* w = (_Float16*)(dst + f16x3_layer_off(F, R, V, l));
        for (int ot = 0; ot < noct; ++ot)
            for (int c = 0; c < nchunks; ++c)
                for (int t = 0; t < 9; ++t)
                    for (int kg = 0; kg < 2; ++kg)
                        for (int o = 0; o < 128; ++o)
                            for (int j = 0; j < 8; ++j) {
                                const int oc = ot * 128 + o, ic = c * 16 + kg * 8 + j;
                                const float v = lsrc[((size_t)oc * F + ic) * 9 + t] * S;
                                const _Float16 hi = (_Float16)v, lo = (_Float16)(v - (float)hi);
                                const size_t stage = ((size_t)ot * nchunks + c) * 3 + t / 3;
                                const size_t base = stage * (W_STAGE / 2) + ((size_t)(t % 3) * 2 + kg) * 2 * 128 * 8;
                                w[base + (size_t)o * 8 + j] = hi;
                                w[base + 128 * 8 + (size_t)o * 8 + j] = lo;
                            }
        lsrc += (size_t)F * F * 9 + F;
    }
}

// The heads over a trunk output in the split layout.
int raz_net_heads_split(const float* W, int F, int R, int V, const unsigned char* trunk, const uint8_t* active, float* policy, float* value,
                        size_t n, hipStream_t s, const uint32_t* list, const uint32_t* n_ptr) {
    hipLaunchKernelGGL(k_heads_split, dim3((unsigned)n), dim3(64), (192 + (size_t)V) * sizeof(float), s, W + heads_off(F, R), trunk, active,
                       policy, value, (int)n, F, V, list, n_ptr);
    return raz_check_launch("raz_net_forward (split heads)");
}

// two activation buffers, then a 64-byte area per row for its range flag (raz_internal.h raz_net_repair_rows): linear in n
size_t raz_net_f16x3_scratch_bytes(int F, size_t n) { return n * ((size_t)2 * F * 256 + RAZ_NET_ROWFLAG_WORDS * 4); }

// F > 256 (a row's f32 activations do not fit a CU's LDS, so there is no in-forward repair): any row out of range raises the sticky flag
__global__ __launch_bounds__(256) void k_flag_unrepaired(const unsigned* __restrict__ rowflag, unsigned* __restrict__ sticky, int n,
                                                         const uint32_t* __restrict__ n_ptr) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int rows = n_ptr ? ((int)*n_ptr < n ? (int)*n_ptr : n) : n;
    if (i < rows && rowflag[(size_t)i * RAZ_NET_ROWFLAG_WORDS]) atomicOr(sticky, 1u);
}
static int raz_net_flag_unrepaired(const unsigned* rowflag, unsigned* sticky, size_t n, const uint32_t* n_ptr, hipStream_t s) {
    hipLaunchKernelGGL(k_flag_unrepaired, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, rowflag, sticky, (int)n, n_ptr);
    return raz_check_launch("raz_net_forward (range flags)");
}

// The sticky range flag lives in the device weight image, after the per-layer scales (raz_net_layout.h leaves 64 floats there).
unsigned* raz_net_f16x3_flag(const float* W, int F, int R, int V) { return (unsigned*)(W + f16x3_scale_off(F, R, V) + (size_t)2 * R + 8); }

// list / n_ptr (both or neither; engine-internal, raz_leaf_cache.hip): evaluate only the exchange rows list[0 .. *n_ptr), packed
// densely in the activation buffers - *n_ptr lives on the device, so every launch keeps its full grid and surplus blocks exit.
int raz_net_forward_f16x3(const float* W, int F, int R, int V, const uint64_t* own, const uint64_t* enemy,
                          const uint8_t* active, float* policy, float* value, size_t n, void* scratch, size_t scratch_bytes,
                          hipStream_t s, const uint32_t* list, const uint32_t* n_ptr) {
    if (!scratch || scratch_bytes < raz_net_f16x3_scratch_bytes(F, n))
        return raz_fail(RAZ_ENOMEM, "raz_net_forward: scratch too small (raz_net_scratch_bytes)");
    unsigned char* bufA = (unsigned char*)scratch;
    unsigned char* bufT = bufA + (size_t)n * F * 256;
    unsigned* sticky = raz_net_f16x3_flag(W, F, R, V);
    unsigned* flag = (unsigned*)(bufT + (size_t)n * F * 256);   // per-row range flags
    const float* scales = W + f16x3_scale_off(F, R, V);
    const auto conv = k_conv3x3_f16x3;
    {   // the kernel's LDS image exceeds the default dynamic limit: raise it once per device
        static unsigned long long attr_devices = 0;   // bit d = done on device d (one process drives one device; a second one still gets its call)
        int dev = 0;
        RAZ_HIP_TRY(hipGetDevice(&dev), "raz_net_forward: hipGetDevice");
        if (dev >= 64 || !(attr_devices >> dev & 1)) {
            RAZ_HIP_TRY(hipFuncSetAttribute((const void*)conv, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES), "raz_net_forward: hipFuncSetAttribute");
            if (dev < 64) attr_devices |= 1ull << dev;
        }
    }
    const unsigned conv_threads = NWAVE * 64;
    hipLaunchKernelGGL(k_conv0_split, dim3((unsigned)n), dim3(256), 0, s, W + conv_off(F, 0), (const raz_bb*)own,
                       (const raz_bb*)enemy, active, bufA, (int)n, F, flag, list, n_ptr);
    const unsigned groups = (unsigned)((n + NWAVE - 1) / NWAVE);
    const unsigned tiles = ((groups + 7) / 8) * 8 * (unsigned)(F / 128);
    const unsigned grid = tiles;
    for (int r = 0; r < R; ++r) {
        const int l1 = 1 + 2 * r, l2 = 2 + 2 * r;
        hipLaunchKernelGGL(conv, dim3(grid), dim3(conv_threads), LDS_BYTES, s,
                           (const unsigned char*)(W + f16x3_layer_off(F, R, V, l1)), W + conv_off(F, l1) + (size_t)F * 9 * F,
                           scales + (l1 - 1), (const unsigned char*)bufA, bufT, (const unsigned char*)nullptr, list ? nullptr : active, (int)n, F, flag, n_ptr RAZ_STAMP_ARG);
        hipLaunchKernelGGL(conv, dim3(grid), dim3(conv_threads), LDS_BYTES, s,
                           (const unsigned char*)(W + f16x3_layer_off(F, R, V, l2)), W + conv_off(F, l2) + (size_t)F * 9 * F,
                           scales + (l2 - 1), (const unsigned char*)bufT, bufA, (const unsigned char*)bufA, list ? nullptr : active, (int)n, F, flag, n_ptr RAZ_STAMP_ARG);
    }
    const int rc = raz_net_heads_split(W, F, R, V, bufA, active, policy, value, n, s, list, n_ptr);
    if (rc != RAZ_OK) return rc;
    // rows whose activations left the f16 range: the same position through the exact-f32 chains (raznet-forward-v1), so a row's
    // answer is a function of its position alone - v2's when it stays in range, v1's when it does not
    if (((size_t)2 * F * 64 + 192 + (size_t)V) * sizeof(float) > 160 * 1024)   // (F > 256: a row does not fit a CU's LDS - not repaired, see raz_net_range_check)
        return raz_net_flag_unrepaired(flag, sticky, n, n_ptr, s);
    return raz_net_repair_rows(W, F, R, V, own, enemy, policy, value, n, flag, sticky, list, n_ptr, s);
}
