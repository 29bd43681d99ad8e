// raz_net_wino.hip — EXPERIMENT, not part of libraz.so (round 4; built only by tools/probe_conv.hip -DPROBE_WINO).  Forward pass of
// WIDE policy/value nets (F % 128 == 0) with the 3x3 convolutions of the trunk as Winograd F(2,3) ALONG THE ROWS on the f16 matrix
// cores with split operands.  RESULT: correct (emulator: <= 1e-5 of the oracle's f32 net on three shapes; hardware: a layer == its
// double-precision restatement to 7e-8, a whole 256x10 forward closer to the f64 graph than raznet-forward-v2: policy 2.6e-6 vs
// 5.6e-6), 1.5x fewer matrix instructions - and NOT faster: 1.25 ms per layer at 8192 positions against v2's 1.23-1.29 ms
// (profiles/r4/conv_wino_*.jsonl).  The accumulators a CU can hold (8 waves x 128 registers) cap a workgroup at 4 positions x 128
// output channels x 4 points, at which the layer moves 8.6 GB from L2 / HBM into the CUs instead of v2's 4.0 GB (the transformed
// weights are 4/3 the bytes and are re-read per 4 positions instead of 8; the transformed activations are twice the bytes): on zero
// operands it runs at ~8.8 TB/s of CU-side traffic, 50 % matrix-pipe busy, and on random operands both kernels sit at the same
// power-limited time.  Kept for the record and for tools/probe_wino (check / time); DESIGN.md 4.4.
//
// Why: the split-operand kernel of raznet-forward-v2 (raz_net_f16x3.hip) is limited by POWER, not by its schedule - on zero
// operands the same instruction stream runs 1.27-1.33x faster (2.4 GHz instead of the 1.75-1.9 GHz the chip holds under random
// operands; profiles/r4/conv_f16x3_probe_session1.jsonl) - so the lever is fewer matrix instructions per result, not a denser
// stream.  Winograd's minimal filtering F(2,3) computes two adjacent outputs of a 3-tap correlation with 4 multiplications
// instead of 6.  Applied along x only (the rows of the 3x3 filter stay direct):
//     inputs  d0..d3 = x[2t-1], x[2t], x[2t+1], x[2t+2]           (t = 0..3: the four 2-wide tiles of a board row, zero padded)
//     V0 = d0 - d2     V1 = d1 + d2     V2 = d2 - d1     V3 = d1 - d3
//     U0 = g0          U1 = (g0 + g1 + g2) / 2           U2 = (g0 - g1 + g2) / 2          U3 = g2          (g = a filter row)
//     M_a[oc][y][t] = sum over ic, dy of U_a[oc][ic][dy] * V_a[ic][y + dy - 1][t]        (4 GEMMs with K = 3 F, N = 32 per position)
//     y[2t] = (M0 + M1) + M2          y[2t+1] = (M1 - M2) - M3
// 4 x 3 = 12 products per output pair instead of 18: 1.5x fewer matrix instructions.  Every product is still the three-instruction
// split of v2 (hi*hi + hi*lo + lo*hi of (hi, lo) = (f16(x), f16(x - hi)), one f32 accumulator), the U_a are formed in double on the
// host and scaled per layer by a power of two, the V_a are formed in f32 by the PRODUCING layer's epilogue and stored split.  An
// accumulator now sums 48 instruction triples instead of 144, which is why the result is no further from the f64 graph than v2's
// (tools/eval_winograd_numerics.py; tests: <= 1e-5 vs fp32 torch like v2).  Not bit-identical to v1 or v2.
//
// Activations in HBM (per position):
//     V  "transformed": [16-channel chunk][plane = point a (4) x k-group of 8 channels (2) x {hi, lo}][col = y * 4 + t (32)][8 halfs]
//                       = 8 KiB per chunk, F * 512 bytes per position: what the conv kernel's LDS image is made of (LDS-DMA)
//     P  "plain":       v2's split layout [chunk][k-group x {hi, lo}][square 64][8 halfs], F * 256 bytes: the skip connection
//                       of a residual block and the heads read it (k_heads_split of raz_net_f16x3.hip, unchanged)
// A block's first convolution reads V(x), writes V(t); its second reads V(t) and P(x), writes V(x') and P(x').
//
// k_conv3x3_wino, GEMM view per point a: D_a[oc, col] += U_a[oc, k] * V_a[k, col], k = (chunk, dy, channel in chunk).
//   workgroup = 8 waves = 128 output channels x 4 positions;
//   wave      = 4 positions x 32 output channels x 2 of the 4 points = 2 x 4 MFMA tiles (128 accumulator registers);
//               waves w and w ^ 4 hold the two point pairs of the same outputs and swap half of their tiles through LDS at the end,
//               so that each finishes 2 positions x 32 channels with all four M_a in registers
//   weights   = NOT staged in LDS: a wave's A operands (its 32 channels x its 2 points x {hi, lo} = 4 KiB per (chunk, dy) stage) are
//               private to it, so they go straight from L2 into registers (4 global_load_dwordx4 of 1 KiB per stage, the layout is
//               the operand order), requested TWO stages ahead.  The first version staged them through LDS one stage ahead with a
//               barrier per stage: a staging round trip (3,100 cycles measured) is twice a stage's matrix time (1,536), so the
//               kernel waited on every stage and was no faster than v2 (profiles/r4/conv_wino_v0_*.jsonl)
//   K loop    = 16 chunks x 3 rows (dy); a chunk's transformed activations of the 4 positions (32 KiB) arrive by LDS-DMA in a ring
//               of three images, requested two chunks ahead: ONE barrier per chunk (72 matrix instructions per wave), raw s_barrier
//               with explicit waits - __syncthreads() would drain the weight loads in flight
//   rows      = the dy shift is +-4 columns of the 32-column plane; off-board rows read the zero block behind the position's planes
//               (per-lane addresses precomputed, no predicates, bank-conflict free like v2's taps)
//   epilogue  = output transform, * 1/S, + bias, (+ skip), relu, then BOTH forms of the result: P (plain, if the layer feeds a
//               skip connection or the heads) and V for the next layer (the neighbouring tiles' border columns come from the
//               adjacent lanes by DPP), each split into (hi, lo), 8-byte stores
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../reversi-alpha-zero_amd/csrc/raz_bitboard.h"
#include "../../reversi-alpha-zero_amd/csrc/raz_detmath.h"
#include "../../reversi-alpha-zero_amd/csrc/raz_internal.h"
#include "../../reversi-alpha-zero_amd/csrc/raz_net_layout.h"

// region-5 offsets of the experiment's weight image (the product's raz_net_layout.h ends with region 4)
RAZ_HD_LAYOUT size_t wino_layer_floats(int F) { return (size_t)F * F * 12; }
RAZ_HD_LAYOUT size_t wino_off(int F, int R, int V) { return (f16x3_scale_off(F, R, V) + (size_t)2 * R + 64 + 63) / 64 * 64; }
RAZ_HD_LAYOUT size_t wino_layer_off(int F, int R, int V, int l) { return wino_off(F, R, V) + (size_t)(l - 1) * wino_layer_floats(F); }  // l >= 1
RAZ_HD_LAYOUT size_t wino_scale_off(int F, int R, int V) { return wino_off(F, R, V) + (size_t)2 * R * wino_layer_floats(F); }

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int OCT = 128;                          // output channels per workgroup
constexpr int NPOS = 4;                           // positions per workgroup
constexpr int NWAVE = 8;
constexpr int W_STAGE = 4 * 4 * 2 * 2 * 32 * 16;  // 32,768 B: [oc quarter 4][point 4][hi/lo][k-group 2][oc 32][16 B]
constexpr int V_POS = 16 * 512;                   // 8,192 B: [plane 16][col 32][16 B] - one (position, chunk)
constexpr int Z_BYTES = 3072;                     // zero block behind a position's planes: off-board reads land at 8192 + slot * 16 + {0, 512, 2048, 2560}
constexpr int L_POS = V_POS + Z_BYTES;            // 11,264 B per position in LDS
constexpr int V_IMG = NPOS * L_POS;               // 45,056 B: one chunk's image
constexpr int NRING = 3;                          // images in flight: the chunk being read + two requested
constexpr int LDS_BYTES = NRING * V_IMG;          // 135,168 B: one workgroup of 8 waves per CU
constexpr int X_WAVE = 16384;                     // epilogue exchange: 64 accumulator registers per wave, 8 x 16 KiB <= LDS_BYTES
constexpr int P_CHUNK = 4 * 64 * 16;              // plain layout: 4,096 B per (position, chunk)

#define GLDS16(gptr, lptr)                                                                                      \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),                     \
                                     (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

__device__ inline float dpp_from_lower_lane(float v) {   // lane i <- lane i - 1 within its row of 16 lanes, 0 at the row's first lane
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
}
__device__ inline float dpp_from_upper_lane(float v) {   // lane i <- lane i + 1 within its row of 16 lanes, 0 at the row's last lane
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x101, 0xf, 0xf, true));
}

// The K loop's memory traffic is issued and awaited BY HAND (inline assembly): the compiler's own wait insertion (a) drains every
// load in flight at __syncthreads() (its release fence), (b) makes any LDS read wait for every LDS-DMA in flight (it cannot tell
// the ring slots apart) and (c) falls back to vmcnt(0) around the loop's back edge - each of which turns a two-stage prefetch into
// none.  Vector loads return in order, so "at most N operations outstanding" means everything older than the N newest has landed:
//   load_weights4   four 1 KiB global_load_dwordx4 (a wave's A operands of one stage) into registers
//   stage_acts4     four 1 KiB LDS-DMA pieces of an activation image
//   wait_loads<N>   s_waitcnt vmcnt(N), tied to the registers it makes valid (the compiler may not move their uses above it)
//   wg_barrier<N>   s_waitcnt vmcnt(N) lgkmcnt(0) + s_barrier
// On the wave emulator (tests/native/wave_emu, synchronous copies) they are plain C++.
#ifdef RAZ_WAVE_EMU
__device__ inline void load_weights4(f32x4 (&w)[4], const unsigned char* p) {
    for (int i = 0; i < 4; ++i) w[i] = *(const f32x4*)(p + i * 1024);
}
__device__ inline void stage_acts4(const unsigned char* g, unsigned char* l) {
    for (int i = 0; i < 4; ++i) GLDS16(g + i * 1024, l + i * 1024);
}
template <int N> __device__ inline void wait_loads(f32x4 (&)[4]) {}
template <int N> __device__ inline void wg_barrier() { __syncthreads(); }
#else
__device__ inline void load_weights4(f32x4 (&w)[4], const unsigned char* p) {
    asm volatile("global_load_dwordx4 %0, %4, off\n\t"
                 "global_load_dwordx4 %1, %4, off offset:1024\n\t"
                 "global_load_dwordx4 %2, %4, off offset:2048\n\t"
                 "global_load_dwordx4 %3, %4, off offset:3072"
                 : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]) : "v"(p) : "memory");
}
__device__ inline void stage_acts4(const unsigned char* g, unsigned char* l) {   // g: per-lane source (+ lane * 16), l: wave-uniform LDS destination
    const uint32_t la = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)l;
    // the instruction's immediate offset advances BOTH addresses (global and LDS): one M0 for the four pieces
    asm volatile("s_mov_b32 m0, %1\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %0, off\n\t"
                 "global_load_lds_dwordx4 %0, off offset:1024\n\t"
                 "global_load_lds_dwordx4 %0, off offset:2048\n\t"
                 "global_load_lds_dwordx4 %0, off offset:3072"
                 :: "v"(g), "s"(la) : "memory", "m0");
}
template <int N> __device__ inline void wait_loads(f32x4 (&w)[4]) {
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]) : "n"(N) : "memory");
}
template <int N> __device__ inline void wg_barrier() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(N) : "memory");
}
#endif

// inV: transformed activations of the layer input.  outV / outP: the result in transformed / plain form (either may be null).
// skipP: plain activations added before the relu (null: none).  Wl: this layer's region-5 weights.
// grid = ceil(n / 4 / 8) * 8 * F / 128, block 512.
__global__ __launch_bounds__(512, 2) void k_conv3x3_wino(const unsigned char* __restrict__ Wl, const float* __restrict__ bias,
                                                         const float* __restrict__ inv_scale_ptr, const unsigned char* inV, unsigned char* outV,
                                                         unsigned char* outP, const unsigned char* skipP, const uint8_t* __restrict__ active, int n,
                                                         int F, unsigned* __restrict__ flag, const uint32_t* __restrict__ n_ptr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (n_ptr) n = (int)*n_ptr < n ? (int)*n_ptr : n;   // rows 0..n-1 of a compacted batch (raz_leaf_cache.hip): the count lives on the device
    const int noct = F / OCT, nchunks = F / 16;
    // blocks b and b + 8 run on the same XCD (round-robin dispatch): give them the oc tiles of the SAME positions
    const int b = blockIdx.x;
    const int ot = (b >> 3) % noct;
    const int pg = (b / (8 * noct)) * 8 + (b & 7);
    const int p0 = pg * NPOS;
    if (p0 >= n) return;
    const int ocq = wv & 3, pq = wv >> 2;   // 32-channel quarter and point pair of this wave
    const size_t posV = (size_t)F * 512, posP = (size_t)F * 256;
    {   // the zero blocks of the three images: 12 x 3,072 B
        for (int k = tid; k < NRING * NPOS * (Z_BYTES / 16); k += NWAVE * 64) {
            const int blk = k / (Z_BYTES / 16), u = k % (Z_BYTES / 16);
            ((f32x4*)(lds + (blk / NPOS) * V_IMG + (blk % NPOS) * L_POS + V_POS))[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
    // per-lane LDS byte offsets (inside an image) of the B operand for the three rows dy: position 0 of the group, this wave's first
    // point, the hi plane; + q * L_POS for position q, + 2048 for the second point, + 512 for the lo plane
    const int kg = lane >> 5, col = lane & 31;
    uint32_t boff[3];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int yy = (col >> 2) + dy - 1, c2 = col + (dy - 1) * 4;
        boff[dy] = (yy >= 0 && yy < 8) ? (uint32_t)(pq * 4096 + kg * 1024 + c2 * 16) : (uint32_t)(V_POS + (c2 & 15) * 16);
    }
    f32x16 acc[2][4];   // [point of the pair][position of the group]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][q][r] = 0.f;
    // this wave's A operands of stage st: 4 KiB at wsrc + st * W_STAGE: [point of the pair][hi, lo][lane][16 B]
    const unsigned char* wsrc = Wl + (size_t)ot * nchunks * 3 * W_STAGE + (size_t)(ocq * 4 + pq * 2) * 2048 + lane * 16;
    // the activation pieces this wave moves: position wv >> 1 of the group, half wv & 1 of its 8 KiB
    const int lpos = p0 + (wv >> 1);
    const unsigned char* asrc = inV + (size_t)(lpos < n ? lpos : n - 1) * posV + (wv & 1) * 4096 + lane * 16;
    auto issue_acts = [&](int c, int slot) {
        stage_acts4(asrc + (size_t)c * V_POS, lds + slot * V_IMG + (wv >> 1) * L_POS + (wv & 1) * 4096);
    };
    f32x4 wreg[3][4];   // [stage % 3][point of the pair * 2 + {hi, lo}]: the weights of three stages, two of them in flight
    const int nstages = nchunks * 3;
    // Requests past the end are clamped to the last chunk / stage instead of skipped (a duplicate into a free slot): the order and
    // number of operations in flight is the same in every iteration, which is what the hand-placed waits count on.  In program order:
    //   DMA(0) DMA(1) W(0) W(1) | per chunk c: [barrier] DMA(c+2) W(3c+2) <stage 3c> W(3c+3) <stage 3c+1> W(3c+4) <stage 3c+2>
    // (4 operations each).  Needed at the barrier of chunk c: DMA(c) - younger ones outstanding: DMA(c+1) W(3c) W(3c+1) = 12;
    // at stage 3c: W(3c) - younger: W(3c+1) DMA(c+2) W(3c+2) = 12; at 3c+1: W(3c+1) - younger: DMA(c+2) W(3c+2) W(3c+3) = 12; at 3c+2: W(3c+2) -
    // younger: W(3c+3) W(3c+4) = 8.
    issue_acts(0, 0);
    issue_acts(nchunks > 1 ? 1 : 0, 1);
    load_weights4(wreg[0], wsrc);
    load_weights4(wreg[1], wsrc + (size_t)(nstages > 1 ? 1 : 0) * W_STAGE);
    for (int c = 0; c < nchunks; ++c) {
        wg_barrier<12>();   // chunk c's image has landed for everybody, and everybody is done reading chunk c - 1's: its slot takes chunk c + 2
        issue_acts(c + 2 < nchunks ? c + 2 : nchunks - 1, (c + 2) % NRING);   // (past the end: a duplicate into the free slot)
        const uint32_t abase = (uint32_t)((c % NRING) * V_IMG);
        // a chunk = 24 units (dy, point a, position q) of three matrix instructions; the B operands of unit u + 1 are requested before
        // the matrix instructions of unit u
        h8 bh[2], bl[2];
        bh[0] = *(const h8*)(lds + abase + boff[0]);
        bl[0] = *(const h8*)(lds + abase + boff[0] + 512);
#pragma unroll
        for (int u = 0; u < 24; ++u) {
            const int dy = u / 8, a = (u % 8) / 4, q = u % 4;
            if (u % 8 == 0) {
                const int st2 = c * 3 + dy + 2;
                load_weights4(wreg[(dy + 2) % 3], wsrc + (size_t)(st2 < nstages ? st2 : nstages - 1) * W_STAGE);
                if (dy == 2) wait_loads<8>(wreg[dy]);
                else wait_loads<12>(wreg[dy]);
            }
            if (u + 1 < 24) {
                const int dy1 = (u + 1) / 8, a1 = ((u + 1) % 8) / 4, q1 = (u + 1) % 4;
                bh[(u + 1) & 1] = *(const h8*)(lds + abase + boff[dy1] + q1 * L_POS + a1 * 2048);
                bl[(u + 1) & 1] = *(const h8*)(lds + abase + boff[dy1] + q1 * L_POS + a1 * 2048 + 512);
            }
            const h8 ah = __builtin_bit_cast(h8, wreg[dy][a * 2]), al = __builtin_bit_cast(h8, wreg[dy][a * 2 + 1]);
            acc[a][q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[u & 1], acc[a][q], 0, 0, 0);
            acc[a][q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[u & 1], acc[a][q], 0, 0, 0);
            acc[a][q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[u & 1], acc[a][q], 0, 0, 0);
        }
    }
    // ---- epilogue.  The point pairs swap tiles: wave pq = 0 finishes positions 0, 1 of the group (needs the partner's M2, M3 of them),
    // wave pq = 1 positions 2, 3 (needs M0, M1).  Each writes the 64 registers it gives away: [tile = a' * 2 + position][register group g of 4][lane][16 B].
    wg_barrier<0>();   // every wave is done with the images, and nothing of its own is in flight any more
    unsigned char* xmine = lds + wv * X_WAVE + lane * 16;
    const unsigned char* xpartner = lds + (wv ^ 4) * X_WAVE + lane * 16;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int qq = 0; qq < 2; ++qq)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x16& give = pq == 0 ? acc[a][2 + qq] : acc[a][qq];
                *(f32x4*)(xmine + ((a * 2 + qq) * 4 + g) * 1024) = (f32x4){give[g * 4], give[g * 4 + 1], give[g * 4 + 2], give[g * 4 + 3]};
            }
    __syncthreads();
    const float inv_scale = *inv_scale_ptr;
    const int t_in_row = col & 3;
    bool over = false;
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) {
        const int pos = p0 + pq * 2 + qq;
        if (!(pos < n && (!active || active[pos]))) continue;   // wave-uniform
        unsigned char* oV = outV ? outV + (size_t)pos * posV : nullptr;
        unsigned char* oP = outP ? outP + (size_t)pos * posP : nullptr;
        const unsigned char* sP = skipP ? skipP + (size_t)pos * posP : nullptr;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            // D layout: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5): register group g = channels 8 g + 4 kg + 0..3 of the tile
            const f32x4 r0 = *(const f32x4*)(xpartner + ((0 * 2 + qq) * 4 + g) * 1024);
            const f32x4 r1 = *(const f32x4*)(xpartner + ((1 * 2 + qq) * 4 + g) * 1024);
            const int oc8 = ot * OCT + ocq * 32 + g * 8;   // this lane pair's 8-channel group; this lane holds 4 of them
            const f32x4 bv = *(const f32x4*)(bias + oc8 + 4 * kg);
            float e[4], o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float m0, m1, m2, m3;
                if (pq == 0) { m0 = acc[0][qq][g * 4 + j]; m1 = acc[1][qq][g * 4 + j]; m2 = r0[j]; m3 = r1[j]; }
                else { m0 = r0[j]; m1 = r1[j]; m2 = acc[0][2 + qq][g * 4 + j]; m3 = acc[1][2 + qq][g * 4 + j]; }
                e[j] = ((m0 + m1) + m2) * inv_scale + bv[j];
                o[j] = ((m1 - m2) - m3) * inv_scale + bv[j];
            }
            const size_t punit = (size_t)(oc8 >> 4) * P_CHUNK + (size_t)((oc8 >> 3) & 1) * 2048 + (size_t)(2 * col) * 16 + kg * 8;
            if (sP) {
                const h4 seh = *(const h4*)(sP + punit), sel = *(const h4*)(sP + punit + 1024);
                const h4 soh = *(const h4*)(sP + punit + 16), sol = *(const h4*)(sP + punit + 16 + 1024);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    e[j] = e[j] + ((float)seh[j] + (float)sel[j]);
                    o[j] = o[j] + ((float)soh[j] + (float)sol[j]);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                e[j] = e[j] > 0.0f ? e[j] : 0.0f;
                o[j] = o[j] > 0.0f ? o[j] : 0.0f;
                over |= !(e[j] + o[j] < 60000.0f);   // V1 = e + o must stay in the f16 range (and so do e, o and the differences)
            }
            if (oP) {
                h4 eh, el, oh4, ol;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    eh[j] = (_Float16)e[j];
                    el[j] = (_Float16)(e[j] - (float)eh[j]);
                    oh4[j] = (_Float16)o[j];
                    ol[j] = (_Float16)(o[j] - (float)oh4[j]);
                }
                *(h4*)(oP + punit) = eh;
                *(h4*)(oP + punit + 1024) = el;
                *(h4*)(oP + punit + 16) = oh4;
                *(h4*)(oP + punit + 16 + 1024) = ol;
            }
            if (oV) {
                const size_t vunit = (size_t)(oc8 >> 4) * V_POS + (size_t)((oc8 >> 3) & 1) * 1024 + (size_t)col * 16 + kg * 8;
                h4 vh[4], vl[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    // x[2t-1] = the odd output of the tile to the left, x[2t+2] = the even output of the tile to the right (0 off the board)
                    float left = dpp_from_lower_lane(o[j]), right = dpp_from_upper_lane(e[j]);
                    left = t_in_row == 0 ? 0.0f : left;
                    right = t_in_row == 3 ? 0.0f : right;
                    const float v[4] = {left - o[j], e[j] + o[j], o[j] - e[j], e[j] - right};
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        vh[a][j] = (_Float16)v[a];
                        vl[a][j] = (_Float16)(v[a] - (float)vh[a][j]);
                    }
                }
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    *(h4*)(oV + vunit + a * 2048) = vh[a];
                    *(h4*)(oV + vunit + a * 2048 + 512) = vl[a];
                }
            }
        }
    }
    if (over) atomicOr(flag, 1u);   // an activation beyond the f16 range: the caller must fall back to the f32 kernel
}

// Layer 0: 2 bit planes -> F channels, exact f32 chains as in k_conv0_split, written in BOTH forms: plain (the first block's
// skip connection) and transformed (the first block's first convolution).  A position's 16-channel chunks are spread over the 4
// waves of a workgroup (lane = square); the even-x lanes form and store their tile's four V points, with the neighbouring
// squares' values fetched across lanes.
__global__ __launch_bounds__(256) void k_conv0_wino(const float* __restrict__ W0, const raz_bb* __restrict__ own,
                                                    const raz_bb* __restrict__ enemy, const uint8_t* __restrict__ active,
                                                    unsigned char* outP, unsigned char* outV, int n, int F, unsigned* __restrict__ flag,
                                                    const uint32_t* __restrict__ list, const uint32_t* __restrict__ n_ptr) {
    const int pos = blockIdx.x, lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: the weight reads below stay scalar loads
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
    unsigned char* op = outP + (size_t)pos * F * 256;
    unsigned char* ov = outV + (size_t)pos * F * 512;
    const int colv = y * 4 + (x >> 1);
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
            h8 hi, lo, vh[4], vl[4];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float r = acc[g * 8 + j] > 0.0f ? acc[g * 8 + j] : 0.0f;
                hi[j] = (_Float16)r;
                lo[j] = (_Float16)(r - (float)hi[j]);
                // this lane as the tile's even square: d0 = x - 1 (0 off the board), d1 = self, d2 = x + 1, d3 = x + 2 (0 off the board)
                const float l1 = __shfl(r, (lane + 63) & 63), r1 = __shfl(r, (lane + 1) & 63), r2 = __shfl(r, (lane + 2) & 63);
                const float d0 = x >= 1 ? l1 : 0.0f, d3 = x + 2 <= 7 ? r2 : 0.0f;
                const float v[4] = {d0 - r1, r + r1, r1 - r, r - d3};
                over |= !(r + r1 < 60000.0f);
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    vh[a][j] = (_Float16)v[a];
                    vl[a][j] = (_Float16)(v[a] - (float)vh[a][j]);
                }
            }
            *(h8*)(op + (size_t)ocb * P_CHUNK + (g * 2 + 0) * 1024 + lane * 16) = hi;
            *(h8*)(op + (size_t)ocb * P_CHUNK + (g * 2 + 1) * 1024 + lane * 16) = lo;
            if ((x & 1) == 0) {
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    *(h8*)(ov + (size_t)ocb * V_POS + (a * 4 + g * 2 + 0) * 512 + colv * 16) = vh[a];
                    *(h8*)(ov + (size_t)ocb * V_POS + (a * 4 + g * 2 + 1) * 512 + colv * 16) = vl[a];
                }
            }
        }
    }
    if (over) atomicOr(flag, 1u);
}

}  // namespace

// Host side of raz_net_load for region 5: `src` = the blob's float parameters, `dst` = the device image being built.
void raz_net_build_wino(const float* src, float* dst, int F, int R, int V) {
    const float* lsrc = src + ((size_t)F * 18 + F);   // layer 1
    float* scales = dst + wino_scale_off(F, R, V);
    const int nchunks = F / 16, noct = F / 128;
    std::vector<float> U((size_t)4 * F * F * 3);      // [a][oc][ic][dy]
    for (int l = 1; l < 2 * R + 1; ++l) {
        float mx = 0.f;
        for (int oc = 0; oc < F; ++oc)
            for (int ic = 0; ic < F; ++ic)
                for (int dy = 0; dy < 3; ++dy) {
                    const float* g = lsrc + ((size_t)oc * F + ic) * 9 + dy * 3;
                    const double u[4] = {(double)g[0], ((double)g[0] + (double)g[1] + (double)g[2]) * 0.5,
                                         ((double)g[0] - (double)g[1] + (double)g[2]) * 0.5, (double)g[2]};
                    for (int a = 0; a < 4; ++a) {
                        const float uf = (float)u[a];
                        U[(((size_t)a * F + oc) * F + ic) * 3 + dy] = uf;
                        mx = fmaxf(mx, fabsf(uf));
                    }
                }
        int e = 0;
        if (mx > 0.f) frexpf(mx, &e);                 // mx = f * 2^e, f in [0.5, 1)  =>  mx * 2^(15 - e) in [2^14, 2^15)
        const float S = ldexpf(1.0f, 15 - e);
        scales[l - 1] = ldexpf(1.0f, e - 15);
        _Float16* w = (_Float16*)(dst + wino_layer_off(F, R, V, l));
        for (int ot = 0; ot < noct; ++ot)
            for (int c = 0; c < nchunks; ++c)
                for (int dy = 0; dy < 3; ++dy) {
                    const size_t stage = ((size_t)ot * nchunks + c) * 3 + dy;
                    for (int a = 0; a < 4; ++a)
                        for (int kg = 0; kg < 2; ++kg)
                            for (int o = 0; o < 128; ++o)
                                for (int j = 0; j < 8; ++j) {
                                    const int oc = ot * 128 + o, ic = c * 16 + kg * 8 + j;
                                    const float v = U[(((size_t)a * F + oc) * F + ic) * 3 + dy] * S;   // exact: S is a power of two
                                    const _Float16 hi = (_Float16)v, lo = (_Float16)(v - (float)hi);
                                    // [oc quarter][point][hi, lo][k-group][oc in quarter][8 halfs]: (quarter, point, hi/lo) = one 1 KiB load of a wave
                                    const size_t unit = ((size_t)(o >> 5) * 4 + a) * 2;
                                    const size_t base = stage * (W_STAGE / 2) + (size_t)(kg * 32 + (o & 31)) * 8 + j;
                                    w[base + (unit + 0) * 512] = hi;
                                    w[base + (unit + 1) * 512] = lo;
                                }
                }
        lsrc += (size_t)F * F * 9 + F;
    }
}

size_t raz_net_wino_scratch_bytes(int F, size_t n) { return n * (size_t)F * (512 + 512 + 256); }

// list / n_ptr as in raz_net_forward_f16x3.  The stem and the heads are exact-f32 chains (k_conv0_wino here, k_heads_split of
// raz_net_f16x3.hip through raz_net_heads_split).
int raz_net_forward_wino(const float* W, int F, int R, int V, const uint64_t* own, const uint64_t* enemy, const uint8_t* active,
                         float* policy, float* value, size_t n, void* scratch, size_t scratch_bytes, hipStream_t s,
                         const uint32_t* list, const uint32_t* n_ptr) {
    if (!scratch || scratch_bytes < raz_net_wino_scratch_bytes(F, n))
        return raz_fail(RAZ_ENOMEM, "raz_net_forward: scratch too small (raz_net_scratch_bytes)");
    unsigned char* bufV0 = (unsigned char*)scratch;
    unsigned char* bufV1 = bufV0 + n * (size_t)F * 512;
    unsigned char* bufP = bufV1 + n * (size_t)F * 512;
    unsigned* flag = raz_net_f16x3_flag(W, F, R, V);
    const float* scales = W + wino_scale_off(F, R, V);
    {   // the kernel's LDS image exceeds the default dynamic limit: raise it once per device
        static unsigned long long attr_devices = 0;
        int dev = 0;
        RAZ_HIP_TRY(hipGetDevice(&dev), "raz_net_forward: hipGetDevice");
        if (dev >= 64 || !(attr_devices >> dev & 1)) {
            RAZ_HIP_TRY(hipFuncSetAttribute((const void*)k_conv3x3_wino, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES),
                        "raz_net_forward: hipFuncSetAttribute");
            if (dev < 64) attr_devices |= 1ull << dev;
        }
    }
    hipLaunchKernelGGL(k_conv0_wino, dim3((unsigned)n), dim3(256), 0, s, W + conv_off(F, 0), (const raz_bb*)own, (const raz_bb*)enemy,
                       active, bufP, bufV0, (int)n, F, flag, list, n_ptr);
    const unsigned groups = (unsigned)((n + NPOS - 1) / NPOS);
    const unsigned grid = ((groups + 7) / 8) * 8 * (unsigned)(F / OCT);
    const uint8_t* act = list ? nullptr : active;
    for (int r = 0; r < R; ++r) {
        const int l1 = 1 + 2 * r, l2 = 2 + 2 * r;
        hipLaunchKernelGGL(k_conv3x3_wino, dim3(grid), dim3(NWAVE * 64), LDS_BYTES, s, (const unsigned char*)(W + wino_layer_off(F, R, V, l1)),
                           W + conv_off(F, l1) + (size_t)F * 9 * F, scales + (l1 - 1), (const unsigned char*)bufV0, bufV1, (unsigned char*)nullptr,
                           (const unsigned char*)nullptr, act, (int)n, F, flag, n_ptr);
        hipLaunchKernelGGL(k_conv3x3_wino, dim3(grid), dim3(NWAVE * 64), LDS_BYTES, s, (const unsigned char*)(W + wino_layer_off(F, R, V, l2)),
                           W + conv_off(F, l2) + (size_t)F * 9 * F, scales + (l2 - 1), (const unsigned char*)bufV1, r + 1 < R ? bufV0 : (unsigned char*)nullptr,
                           bufP, (const unsigned char*)bufP, act, (int)n, F, flag, n_ptr);
    }
    return raz_net_heads_split(W, F, R, V, bufP, active, policy, value, n, s, list, n_ptr);
}
