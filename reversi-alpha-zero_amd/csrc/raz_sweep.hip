// raz_sweep.hip — batched bitboard sweeps for gfx950: one board per lane over SoA arrays in HBM.
//
// These are the reference's per-position integer ops (lib/bitboard.py find_correct_moves /
// calc_flip, env/reversi_env.py step / _game_over) applied to n independent positions.  They are
// pure streaming integer kernels: ~150 VALU ops against 24-45 bytes per board, so the bound is
// HBM bandwidth, not ALU (no MFMA, no LDS: there is no reuse to stage).  Layout choices:
//   * SoA u64 arrays so that a wave's 64 lanes read 64 consecutive boards: every load/store is
//     fully coalesced;
//   * each thread owns TWO adjacent boards so the u64 streams move as 16 B/lane
//     (global_load_dwordx4, 1 KiB per wave instruction) and the u8 streams as 2 B/lane;
//   * grid = min(needed, 2048) workgroups of 256 threads, grid-stride loop — 8 workgroups per CU
//     on 256 CUs, consecutive workgroups land on consecutive XCDs and there is no inter-block
//     sharing, so no XCD remap is needed.
#include <hip/hip_runtime.h>
#include "raz_bitboard.h"
#include "raz_internal.h"

namespace {

constexpr int kBlock = 256;
constexpr int kMaxGrid = 2048;

inline unsigned grid_for(size_t work_items) {
    size_t g = (work_items + kBlock - 1) / kBlock;
    if (g > (size_t)kMaxGrid) g = kMaxGrid;
    if (g == 0) g = 1;
    return (unsigned)g;
}

__global__ __launch_bounds__(kBlock) void k_legal_moves(const ulonglong2* __restrict__ own2,
                                                        const ulonglong2* __restrict__ enemy2,
                                                        ulonglong2* __restrict__ legal2,
                                                        const raz_bb* __restrict__ own,
                                                        const raz_bb* __restrict__ enemy,
                                                        raz_bb* __restrict__ legal, size_t n) {
    const size_t pairs = n >> 1;
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < pairs; i += stride) {
        ulonglong2 o = own2[i], e = enemy2[i], r;
        r.x = bb_legal_moves(o.x, e.x);
        r.y = bb_legal_moves(o.y, e.y);
        legal2[i] = r;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0)
        legal[n - 1] = bb_legal_moves(own[n - 1], enemy[n - 1]);
}

__global__ __launch_bounds__(kBlock) void k_calc_flip(const uint8_t* __restrict__ pos,
                                                      const raz_bb* __restrict__ own,
                                                      const raz_bb* __restrict__ enemy,
                                                      raz_bb* __restrict__ flipped, size_t n) {
    const size_t pairs = n >> 1;
    const size_t stride = (size_t)gridDim.x * kBlock;
    const ulonglong2* own2 = (const ulonglong2*)own;
    const ulonglong2* enemy2 = (const ulonglong2*)enemy;
    const uchar2* pos2 = (const uchar2*)pos;
    ulonglong2* out2 = (ulonglong2*)flipped;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < pairs; i += stride) {
        ulonglong2 o = own2[i], e = enemy2[i], r;
        uchar2 p = pos2[i];
        r.x = p.x < 64 ? bb_calc_flip(p.x, o.x, e.x) : 0;
        r.y = p.y < 64 ? bb_calc_flip(p.y, o.y, e.y) : 0;
        out2[i] = r;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        uint8_t p = pos[n - 1];
        flipped[n - 1] = p < 64 ? bb_calc_flip(p, own[n - 1], enemy[n - 1]) : 0;
    }
}

__device__ __forceinline__ void step_one(raz_bb& b, raz_bb& w, uint8_t& pl, uint8_t& st, raz_bb& lg,
                                         uint8_t act) {
    if (st != 0) {
        lg = 0;
        return;
    }
    raz_step_result r = bb_env_step(b, w, pl, act);
    b = r.black;
    w = r.white;
    pl = r.player;
    st = r.status;
    lg = r.legal;
}

// Four adjacent boards per thread: the u64 streams move as 2 x 16 B per lane and the u8 streams as
// 4 B per lane, and the four independent flip/mobility computations give the VALU instruction-level
// parallelism (the kernel sits near the crossover between the HBM and the integer-VALU bound).
__global__ __launch_bounds__(kBlock) void k_step(raz_bb* __restrict__ black,
                                                 raz_bb* __restrict__ white,
                                                 uint8_t* __restrict__ player,
                                                 uint8_t* __restrict__ status,
                                                 raz_bb* __restrict__ legal,
                                                 const uint8_t* __restrict__ action, size_t n) {
    const size_t quads = n >> 2;
    const size_t stride = (size_t)gridDim.x * kBlock;
    ulonglong2* black2 = (ulonglong2*)black;
    ulonglong2* white2 = (ulonglong2*)white;
    ulonglong2* legal2 = (ulonglong2*)legal;
    uchar4* player4 = (uchar4*)player;
    uchar4* status4 = (uchar4*)status;
    const uchar4* action4 = (const uchar4*)action;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < quads; i += stride) {
        ulonglong2 b0 = black2[2 * i], b1 = black2[2 * i + 1], w0 = white2[2 * i], w1 = white2[2 * i + 1], l0, l1;
        uchar4 p = player4[i], s = status4[i], a = action4[i];
        step_one(b0.x, w0.x, p.x, s.x, l0.x, a.x);
        step_one(b0.y, w0.y, p.y, s.y, l0.y, a.y);
        step_one(b1.x, w1.x, p.z, s.z, l1.x, a.z);
        step_one(b1.y, w1.y, p.w, s.w, l1.y, a.w);
        black2[2 * i] = b0;
        black2[2 * i + 1] = b1;
        white2[2 * i] = w0;
        white2[2 * i + 1] = w1;
        legal2[2 * i] = l0;
        legal2[2 * i + 1] = l1;
        player4[i] = p;
        status4[i] = s;
    }
    // ragged tail (n % 4 boards)
    const size_t tail0 = quads << 2;
    if (blockIdx.x == 0 && threadIdx.x < (n - tail0)) {
        const size_t i = tail0 + threadIdx.x;
        raz_bb b = black[i], w = white[i], l;
        uint8_t p = player[i], s = status[i];
        step_one(b, w, p, s, l, action[i]);
        black[i] = b;
        white[i] = w;
        legal[i] = l;
        player[i] = p;
        status[i] = s;
    }
}

__device__ __forceinline__ void score_one(raz_bb b, raz_bb w, uint8_t& win, int8_t& diff) {
    int nb = bb_popcount(b), nw = bb_popcount(w);
    win = (uint8_t)(nb > nw ? RAZ_WIN_BLACK : (nb < nw ? RAZ_WIN_WHITE : RAZ_WIN_DRAW));
    diff = (int8_t)(nb - nw);
}

__global__ __launch_bounds__(kBlock) void k_score(const raz_bb* __restrict__ black,
                                                  const raz_bb* __restrict__ white,
                                                  uint8_t* __restrict__ winner,
                                                  int8_t* __restrict__ diff, size_t n) {
    const size_t pairs = n >> 1;
    const size_t stride = (size_t)gridDim.x * kBlock;
    const ulonglong2* black2 = (const ulonglong2*)black;
    const ulonglong2* white2 = (const ulonglong2*)white;
    uchar2* winner2 = (uchar2*)winner;
    char2* diff2 = (char2*)diff;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < pairs; i += stride) {
        ulonglong2 b = black2[i], w = white2[i];
        uchar2 win;
        int8_t d0, d1;
        score_one(b.x, w.x, win.x, d0);
        score_one(b.y, w.y, win.y, d1);
        winner2[i] = win;
        diff2[i] = make_char2(d0, d1);
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0)
        score_one(black[n - 1], white[n - 1], winner[n - 1], diff[n - 1]);
}

__global__ __launch_bounds__(kBlock) void k_d4(const raz_bb* __restrict__ in,
                                               raz_bb* __restrict__ out,
                                               const uint8_t* __restrict__ sym, size_t n) {
    const size_t pairs = n >> 1;
    const size_t stride = (size_t)gridDim.x * kBlock;
    const ulonglong2* in2 = (const ulonglong2*)in;
    ulonglong2* out2 = (ulonglong2*)out;
    const uchar2* sym2 = (const uchar2*)sym;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < pairs; i += stride) {
        ulonglong2 x = in2[i];
        uchar2 s = sym2[i];
        x.x = bb_d4_apply(x.x, (s.x >> 2) & 1, s.x & 3);
        x.y = bb_d4_apply(x.y, (s.y >> 2) & 1, s.y & 3);
        out2[i] = x;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        uint8_t s = sym[n - 1];
        out[n - 1] = bb_d4_apply(in[n - 1], (s >> 2) & 1, s & 3);
    }
}

// planes: one wave writes one position's 2x64 floats per iteration pair; lane = square, so the
// 512-byte row of a position is two fully coalesced 256-byte stores.
__global__ __launch_bounds__(kBlock) void k_planes(const raz_bb* __restrict__ own,
                                                   const raz_bb* __restrict__ enemy,
                                                   float* __restrict__ planes, size_t n) {
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * kBlock) >> 6;
    for (size_t i = wave; i < n; i += nwaves) {
        raz_bb o = own[i], e = enemy[i];
        float* dst = planes + i * 128;
        dst[lane] = (float)((o >> lane) & 1);
        dst[64 + lane] = (float)((e >> lane) & 1);
    }
}

__device__ __forceinline__ uint8_t kth_set_bit(raz_bb m, uint32_t rnd) {
    int c = bb_popcount(m);
    if (c == 0) return RAZ_ACTION_RESIGN;
    int k = (int)(rnd % (uint32_t)c);
    for (int j = 0; j < k; ++j) m &= m - 1;  // drop the k lowest set bits
    return (uint8_t)(__ffsll((long long)m) - 1);
}

__global__ __launch_bounds__(kBlock) void k_pick_kth(const raz_bb* __restrict__ legal,
                                                     const uint32_t* __restrict__ rnd,
                                                     uint8_t* __restrict__ action, size_t n) {
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride)
        action[i] = kth_set_bit(legal[i], rnd[i]);
}

inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

#define RAZ_REQUIRE(cond, msg)                \
    do {                                      \
        if (!(cond)) return raz_fail(RAZ_EINVAL, msg); \
    } while (0)

extern "C" int raz_legal_moves_batch(const uint64_t* own, const uint64_t* enemy, uint64_t* legal,
                                     size_t n, raz_stream_t stream) {
    if (n == 0) return RAZ_OK;
    RAZ_REQUIRE(own && enemy && legal, "raz_legal_moves_batch: NULL array");
    RAZ_REQUIRE(aligned16(own) && aligned16(enemy) && aligned16(legal),
                "raz_legal_moves_batch: arrays must be 16-byte aligned");
    hipLaunchKernelGGL(k_legal_moves, dim3(grid_for(n / 2 + 1)), dim3(kBlock), 0, (hipStream_t)stream,
                       (const ulonglong2*)own, (const ulonglong2*)enemy, (ulonglong2*)legal,
                       (const raz_bb*)own, (const raz_bb*)enemy, (raz_bb*)legal, n);
    return raz_check_launch("raz_legal_moves_batch");
}

extern "C" int raz_calc_flip_batch(const uint8_t* pos, const uint64_t* own, const uint64_t* enemy,
                                   uint64_t* flipped, size_t n, raz_stream_t stream) {
    if (n == 0) return RAZ_OK;
    RAZ_REQUIRE(pos && own && enemy && flipped, "raz_calc_flip_batch: NULL array");
    RAZ_REQUIRE(aligned16(own) && aligned16(enemy) && aligned16(flipped) && ((uintptr_t)pos & 1) == 0,
                "raz_calc_flip_batch: u64 arrays must be 16-byte aligned, pos 2-byte aligned");
    hipLaunchKernelGGL(k_calc_flip, dim3(grid_for(n / 2 + 1)), dim3(kBlock), 0, (hipStream_t)stream,
                       pos, (const raz_bb*)own, (const raz_bb*)enemy, (raz_bb*)flipped, n);
    return raz_check_launch("raz_calc_flip_batch");
}

extern "C" int raz_step_batch(uint64_t* black, uint64_t* white, uint8_t* player, uint8_t* status,
                              uint64_t* legal, const uint8_t* action, size_t n,
                              raz_stream_t stream) {
    if (n == 0) return RAZ_OK;
    RAZ_REQUIRE(black && white && player && status && legal && action, "raz_step_batch: NULL array");
    RAZ_REQUIRE(aligned16(black) && aligned16(white) && aligned16(legal),
                "raz_step_batch: u64 arrays must be 16-byte aligned");
    RAZ_REQUIRE((((uintptr_t)player | (uintptr_t)status | (uintptr_t)action) & 3) == 0,
                "raz_step_batch: u8 arrays must be 4-byte aligned");
    hipLaunchKernelGGL(k_step, dim3(grid_for(n / 4 + 1)), dim3(kBlock), 0, (hipStream_t)stream,
                       (raz_bb*)black, (raz_bb*)white, player, status, (raz_bb*)legal, action, n);
    return raz_check_launch("raz_step_batch");
}

extern "C" int raz_score_batch(const uint64_t* black, const uint64_t* white, uint8_t* winner,
                               int8_t* diff, size_t n, raz_stream_t stream) {
    if (n == 0) return RAZ_OK;
    RAZ_REQUIRE(black && white && winner && diff, "raz_score_batch: NULL array");
    RAZ_REQUIRE(aligned16(black) && aligned16(white), "raz_score_batch: u64 arrays must be 16-byte aligned");
    RAZ_REQUIRE((((uintptr_t)winner | (uintptr_t)diff) & 1) == 0,
                "raz_score_batch: u8 arrays must be 2-byte aligned");
    hipLaunchKernelGGL(k_score, dim3(grid_for(n / 2 + 1)), dim3(kBlock), 0, (hipStream_t)stream,
                       (const raz_bb*)black, (const raz_bb*)white, winner, diff, n);
    return raz_check_launch("raz_score_batch");
}

extern "C" int raz_d4_batch(const uint64_t* in, uint64_t* out, const uint8_t* sym, size_t n,
                            raz_stream_t stream) {
    if (n == 0) return RAZ_OK;
    RAZ_REQUIRE(in && out && sym, "raz_d4_batch: NULL array");
    RAZ_REQUIRE(aligned16(in) && aligned16(out) && ((uintptr_t)sym & 1) == 0,
                "raz_d4_batch: u64 arrays must be 16-byte aligned, sym 2-byte aligned");
    hipLaunchKernelGGL(k_d4, dim3(grid_for(n / 2 + 1)), dim3(kBlock), 0, (hipStream_t)stream,
                       (const raz_bb*)in, (raz_bb*)out, sym, n);
    return raz_check_launch("raz_d4_batch");
}

extern "C" int raz_planes_batch(const uint64_t* own, const uint64_t* enemy, float* planes, size_t n,
                                raz_stream_t stream) {
    if (n == 0) return RAZ_OK;
    RAZ_REQUIRE(own && enemy && planes, "raz_planes_batch: NULL array");
    hipLaunchKernelGGL(k_planes, dim3(grid_for(n * 64)), dim3(kBlock), 0, (hipStream_t)stream,
                       (const raz_bb*)own, (const raz_bb*)enemy, planes, n);
    return raz_check_launch("raz_planes_batch");
}

extern "C" int raz_pick_kth_legal_batch(const uint64_t* legal, const uint32_t* rnd, uint8_t* action,
                                        size_t n, raz_stream_t stream) {
    if (n == 0) return RAZ_OK;
    RAZ_REQUIRE(legal && rnd && action, "raz_pick_kth_legal_batch: NULL array");
    hipLaunchKernelGGL(k_pick_kth, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream,
                       (const raz_bb*)legal, rnd, action, n);
    return raz_check_launch("raz_pick_kth_legal_batch");
}
