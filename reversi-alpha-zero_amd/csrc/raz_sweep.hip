// raz_sweep.hip — batched bitboard sweeps for gfx950: one board per lane over SoA arrays in HBM.
//
// These are the reference's per-position integer ops (lib/bitboard.py find_correct_moves /
// calc_flip, env/reversi_env.py step / _game_over) applied to n independent positions.  They are
// pure streaming integer kernels (no MFMA, no LDS: there is no reuse to stage) - but with a board
// per lane every u64 op costs vector instructions, and it is VALU issue, not HBM, that bounds
// k_step / k_legal_moves: hence the VALU-shaped primitives of raz_bitboard_valu.h.  Layout choices:
//   * SoA u64 arrays so that a wave's 64 lanes read 64 consecutive boards: every load/store is
//     fully coalesced;
//   * each thread owns TWO adjacent boards so the u64 streams move as 16 B/lane
//     (global_load_dwordx4, 1 KiB per wave instruction) and the u8 streams as 2 B/lane;
//   * grid = one workgroup of 256 threads per 1024 boards (k_step, k_legal_moves: one 256-board block per wave), capped
//     at 65536 workgroups; the grid-stride loops only matter beyond that.  Measured on 2^24 boards
//     (profiles/r2/sweep_grid_tuning.txt): caps of 512 / 2048 / 4096 / 16384 workgroups give k_step 0.193 / 0.167 / 0.161 /
//     0.150 ms and k_legal_moves 0.088 / 0.084 / 0.082 / 0.076 ms - many short workgroups beat few persistent ones with
//     software prefetch (the hardware dispatcher overlaps the next workgroup's loads with the current one's arithmetic).
//     Consecutive workgroups land on consecutive XCDs and there is no inter-block sharing, so no XCD remap is needed.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "raz_bitboard.h"
#include "raz_bitboard_valu.h"
#include "raz_sweep_sliced.h"
#include "raz_internal.h"

namespace {

constexpr int kBlock = 256;
constexpr int kMaxGrid = 65536;

inline unsigned grid_for(size_t work_items) {
    size_t g = (work_items + kBlock - 1) / kBlock;
    if (g > (size_t)kMaxGrid) g = kMaxGrid;
    if (g == 0) g = 1;
    return (unsigned)g;
}

// A wave works on blocks of 256 adjacent boards (128 pairs): lane l takes pairs l and 64 + l of the block, so that every
// load / store instruction of the wave covers ONE contiguous 1 KiB run, and the next block's loads are issued before the
// current block is computed (the integer work of a block then overlaps the memory time of the next instead of adding to it).
__global__ __launch_bounds__(kBlock) void k_legal_moves(const ulonglong2* __restrict__ own2,
                                                        const ulonglong2* __restrict__ enemy2,
                                                        ulonglong2* __restrict__ legal2,
                                                        const raz_bb* __restrict__ own,
                                                        const raz_bb* __restrict__ enemy,
                                                        raz_bb* __restrict__ legal, size_t n) {
    const size_t pairs = n >> 1, nblocks = pairs >> 7;
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * kBlock + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * kBlock) >> 6;
    size_t blk = wave;
    ulonglong2 o0, o1, e0, e1;
    if (blk < nblocks) {
        const size_t i = (blk << 7) + lane;
        o0 = own2[i]; o1 = own2[i + 64]; e0 = enemy2[i]; e1 = enemy2[i + 64];
    }
    while (blk < nblocks) {
        const size_t i = (blk << 7) + lane, nb = blk + nwaves;
        ulonglong2 no0, no1, ne0, ne1;
        if (nb < nblocks) {
            const size_t k = (nb << 7) + lane;
            no0 = own2[k]; no1 = own2[k + 64]; ne0 = enemy2[k]; ne1 = enemy2[k + 64];
        }
        ulonglong2 r0, r1;
        r0.x = bbv_legal_moves(o0.x, e0.x);
        r0.y = bbv_legal_moves(o0.y, e0.y);
        r1.x = bbv_legal_moves(o1.x, e1.x);
        r1.y = bbv_legal_moves(o1.y, e1.y);
        legal2[i] = r0;
        legal2[i + 64] = r1;
        o0 = no0; o1 = no1; e0 = ne0; e1 = ne1;
        blk = nb;
    }
    // ragged tail: the boards after the last full block
    for (size_t t = (nblocks << 8) + (size_t)blockIdx.x * kBlock + threadIdx.x; t < n; t += (size_t)gridDim.x * kBlock)
        legal[t] = bbv_legal_moves(own[t], enemy[t]);
}

// find_correct_moves in bit-sliced form (raz_sweep_sliced.h): one single-wave workgroup per superblock of 2048 boards, 32 boards per
// lane.  ~70 vector instructions per board (three transposes + the walks) instead of 134: the kernel is HBM-bound.
__global__ __launch_bounds__(64) void k_legal_moves_sliced(const ulonglong2* __restrict__ own2, const ulonglong2* __restrict__ enemy2,
                                                           ulonglong2* __restrict__ legal2, size_t nsb) {
    for (size_t sb = blockIdx.x; sb < nsb; sb += gridDim.x) {
        const size_t at = sb * 1024 + threadIdx.x;
        uint32_t o[64], e[64], L[64];
        sl::load_boards(own2, at, o);
        sl::load_boards(enemy2, at, e);
        sl::transpose_boards(o);
        sl::transpose_boards(e);
        sl::mobility(o, e, L);
        sl::transpose_boards(L);
        sl::store_boards(legal2, at, L);
    }
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
        r.x = p.x < 64 ? bbv_calc_flip(p.x, o.x, e.x) : 0;
        r.y = p.y < 64 ? bbv_calc_flip(p.y, o.y, e.y) : 0;
        out2[i] = r;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        uint8_t p = pos[n - 1];
        flipped[n - 1] = p < 64 ? bbv_calc_flip(p, own[n - 1], enemy[n - 1]) : 0;
    }
}

// ReversiEnv.step on one board, first part: returns true when the opponent has no move after it - the board then
// still needs step_rest (the mover's own mobility, else the final count).  That case is rare (a pass or the end of a
// game: ~2 % of the steps of a game), so k_step does it out of line for the few boards concerned.
__device__ __forceinline__ bool step_first(raz_bb& b, raz_bb& w, uint8_t& pl, uint8_t& st, raz_bb& lg, uint8_t act) {
    if (st != 0) {
        lg = 0;
        return false;
    }
    raz_step_result r = bbv_env_step_first(b, w, pl, act);
    b = r.black;
    w = r.white;
    pl = r.player;
    lg = r.legal;
    const bool stuck = r.status == RAZ_STEP_OPP_STUCK;
    st = stuck ? 0 : r.status;
    return stuck;
}
__device__ __forceinline__ void step_rest(raz_bb b, raz_bb w, uint8_t pl, uint8_t& st, raz_bb& lg) {
    raz_step_result r;
    r.black = b;
    r.white = w;
    r.player = pl;
    r.legal = 0;
    r.status = RAZ_STEP_OPP_STUCK;
    bbv_env_step_finish(r);
    st = r.status;
    lg = r.legal;
}
__device__ __forceinline__ void step_one(raz_bb& b, raz_bb& w, uint8_t& pl, uint8_t& st, raz_bb& lg, uint8_t act) {
    if (step_first(b, w, pl, st, lg, act)) step_rest(b, w, pl, st, lg);
}

// Blocks of 256 adjacent boards per wave as in k_legal_moves (lane l: pairs l and 64 + l), next block prefetched: four
// independent flip / mobility computations per lane give the VALU instruction-level parallelism, every memory instruction
// of a wave covers one contiguous run (1 KiB for the u64 streams, 128 B for the u8 streams).
struct StepRegs {
    ulonglong2 b0, b1, w0, w1;
    uchar2 p0, p1, s0, s1, a0, a1;
};
__device__ __forceinline__ StepRegs step_load(const ulonglong2* black2, const ulonglong2* white2, const uchar2* player2,
                                              const uchar2* status2, const uchar2* action2, size_t i) {
    StepRegs r;
    r.b0 = black2[i]; r.b1 = black2[i + 64]; r.w0 = white2[i]; r.w1 = white2[i + 64];
    r.p0 = player2[i]; r.p1 = player2[i + 64]; r.s0 = status2[i]; r.s1 = status2[i + 64];
    r.a0 = action2[i]; r.a1 = action2[i + 64];
    return r;
}
__global__ __launch_bounds__(kBlock) void k_step(raz_bb* __restrict__ black,
                                                 raz_bb* __restrict__ white,
                                                 uint8_t* __restrict__ player,
                                                 uint8_t* __restrict__ status,
                                                 raz_bb* __restrict__ legal,
                                                 const uint8_t* __restrict__ action, size_t n) {
    const size_t pairs = n >> 1, nblocks = pairs >> 7;
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * kBlock + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * kBlock) >> 6;
    ulonglong2* black2 = (ulonglong2*)black;
    ulonglong2* white2 = (ulonglong2*)white;
    ulonglong2* legal2 = (ulonglong2*)legal;
    uchar2* player2 = (uchar2*)player;
    uchar2* status2 = (uchar2*)status;
    const uchar2* action2 = (const uchar2*)action;
    size_t blk = wave;
    StepRegs c;
    if (blk < nblocks) c = step_load(black2, white2, player2, status2, action2, (blk << 7) + lane);
    while (blk < nblocks) {
        const size_t i = (blk << 7) + lane, nb = blk + nwaves;
        StepRegs nx;
        if (nb < nblocks) nx = step_load(black2, white2, player2, status2, action2, (nb << 7) + lane);
        ulonglong2 l0, l1;
        unsigned stuck = (unsigned)step_first(c.b0.x, c.w0.x, c.p0.x, c.s0.x, l0.x, c.a0.x);
        stuck |= (unsigned)step_first(c.b0.y, c.w0.y, c.p0.y, c.s0.y, l0.y, c.a0.y) << 1;
        stuck |= (unsigned)step_first(c.b1.x, c.w1.x, c.p1.x, c.s1.x, l1.x, c.a1.x) << 2;
        stuck |= (unsigned)step_first(c.b1.y, c.w1.y, c.p1.y, c.s1.y, l1.y, c.a1.y) << 3;
        // the rare second half, once per lane and round for the lane's lowest board that needs it (a wave's 256 boards
        // hold ~6 such boards: one round almost always, instead of a second mobility computation on every board)
        while (__any(stuck != 0)) {
            if (stuck) {
                const unsigned k = __ffs(stuck) - 1;
                const raz_bb b = k == 0 ? c.b0.x : (k == 1 ? c.b0.y : (k == 2 ? c.b1.x : c.b1.y));
                const raz_bb w = k == 0 ? c.w0.x : (k == 1 ? c.w0.y : (k == 2 ? c.w1.x : c.w1.y));
                const uint8_t pl = k == 0 ? c.p0.x : (k == 1 ? c.p0.y : (k == 2 ? c.p1.x : c.p1.y));
                uint8_t st;
                raz_bb lg;
                step_rest(b, w, pl, st, lg);
                if (k == 0) { c.s0.x = st; l0.x = lg; }
                else if (k == 1) { c.s0.y = st; l0.y = lg; }
                else if (k == 2) { c.s1.x = st; l1.x = lg; }
                else { c.s1.y = st; l1.y = lg; }
                stuck &= stuck - 1;
            }
        }
        black2[i] = c.b0; black2[i + 64] = c.b1;
        white2[i] = c.w0; white2[i + 64] = c.w1;
        legal2[i] = l0; legal2[i + 64] = l1;
        player2[i] = c.p0; player2[i + 64] = c.p1;
        status2[i] = c.s0; status2[i + 64] = c.s1;
        c = nx;
        blk = nb;
    }
    // ragged tail: the boards after the last full block
    for (size_t t = (nblocks << 8) + (size_t)blockIdx.x * kBlock + threadIdx.x; t < n; t += (size_t)gridDim.x * kBlock) {
        raz_bb b = black[t], w = white[t], l;
        uint8_t p = player[t], st = status[t];
        step_one(b, w, p, st, l, action[t]);
        black[t] = b;
        white[t] = w;
        legal[t] = l;
        player[t] = p;
        status[t] = st;
    }
}

// ReversiEnv.step in bit-sliced form (raz_sweep_sliced.h step_boards): one single-wave workgroup per superblock of 2048 boards.
__global__ __launch_bounds__(64) void k_step_sliced(ulonglong2* __restrict__ black2, ulonglong2* __restrict__ white2, uchar2* __restrict__ player2,
                                                    uchar2* __restrict__ status2, ulonglong2* __restrict__ legal2, const uchar2* __restrict__ action2,
                                                    size_t nsb) {
    for (size_t sb = blockIdx.x; sb < nsb; sb += gridDim.x) {
        const size_t at = sb * 1024 + threadIdx.x;
        uint32_t b[64], w[64], L[64], pw[8], sw[8], aw[8];
        sl::load_boards(black2, at, b);
        sl::load_boards(white2, at, w);
        sl::sfor<8>([&](auto q_) __attribute__((always_inline)) {   // rows 2q, 2q + 1 = the lane's boards 4q .. 4q + 3
            constexpr int q = decltype(q_)::value;
            const uchar2 p0 = player2[at + (size_t)(2 * q) * 64], p1 = player2[at + (size_t)(2 * q + 1) * 64];
            const uchar2 s0 = status2[at + (size_t)(2 * q) * 64], s1 = status2[at + (size_t)(2 * q + 1) * 64];
            const uchar2 a0 = action2[at + (size_t)(2 * q) * 64], a1 = action2[at + (size_t)(2 * q + 1) * 64];
            pw[q] = (uint32_t)p0.x | ((uint32_t)p0.y << 8) | ((uint32_t)p1.x << 16) | ((uint32_t)p1.y << 24);
            sw[q] = (uint32_t)s0.x | ((uint32_t)s0.y << 8) | ((uint32_t)s1.x << 16) | ((uint32_t)s1.y << 24);
            aw[q] = (uint32_t)a0.x | ((uint32_t)a0.y << 8) | ((uint32_t)a1.x << 16) | ((uint32_t)a1.y << 24);
        });
        sl::transpose_boards(b);
        sl::transpose_boards(w);
        RAZ_SL_PHASE();
        const sl::StepMasks m = sl::step_boards(b, w, L, pw, sw, aw);
        RAZ_SL_PHASE();
        if (m.overlap) {
            // one of this lane's boards has a square that is black AND white - no position of the game, and the one input on which the
            // reference's ray arithmetic is not a walk along the board's lines (raz_sweep_sliced.h StepMasks): this lane's 32 boards are
            // stepped one by one with the reference-shaped primitives instead, from memory (nothing of them has been stored yet)
            raz_bb* bl = (raz_bb*)black2;
            raz_bb* wh = (raz_bb*)white2;
            raz_bb* lg = (raz_bb*)legal2;
            uint8_t* pl = (uint8_t*)player2;
            uint8_t* st = (uint8_t*)status2;
            const uint8_t* ac = (const uint8_t*)action2;
            for (int k = 0; k < 32; ++k) {
                const size_t t = 2 * (at + (size_t)(k >> 1) * 64) + (size_t)(k & 1);
                raz_bb bb_ = bl[t], ww_ = wh[t], ll_;
                uint8_t pp_ = pl[t], ss_ = st[t];
                step_one(bb_, ww_, pp_, ss_, ll_, ac[t]);
                bl[t] = bb_; wh[t] = ww_; lg[t] = ll_; pl[t] = pp_; st[t] = ss_;
            }
            continue;
        }
        sl::transpose_boards(b);
        sl::transpose_boards(w);
        sl::transpose_boards(L);
        sl::store_boards(black2, at, b);
        sl::store_boards(white2, at, w);
        sl::store_boards(legal2, at, L);
        // player / status of every board (bb_env_step's cases; the boards are in memory layout again: b[k] / b[32 + k] = the halves of board k)
        uint32_t npw[8], nsw[8];
        sl::sfor<8>([&](auto q_) __attribute__((always_inline)) {
            constexpr int q = decltype(q_)::value;
            uint32_t np = 0, ns = 0;
            sl::sfor<4>([&](auto i_) __attribute__((always_inline)) {
                constexpr int i = decltype(i_)::value, k = 4 * q + i;
                const uint32_t p = (pw[q] >> (8 * i)) & 0xffu, st = (sw[q] >> (8 * i)) & 0xffu, a = (aw[q] >> (8 * i)) & 0xffu;
                const uint32_t other_wins = p == RAZ_PLAYER_BLACK ? RAZ_WIN_WHITE : RAZ_WIN_BLACK;
                const bool moved = (m.moved >> k) & 1u, nz1 = (m.nz1 >> k) & 1u, nz2 = (m.nz2 >> k) & 1u;
                uint32_t p2 = p, s2 = st;
                if (st == 0) {
                    if (a == RAZ_ACTION_RESIGN) s2 = other_wins | RAZ_STATUS_RESIGNED;
                    else if (!moved) s2 = other_wins | RAZ_STATUS_ILLEGAL;
                    else if (nz1) p2 = (3u - p) & 0xffu;
                    else if (!nz2) {
                        const int nb = __builtin_popcount(b[k]) + __builtin_popcount(b[32 + k]), nw = __builtin_popcount(w[k]) + __builtin_popcount(w[32 + k]);
                        s2 = nb > nw ? RAZ_WIN_BLACK : (nb < nw ? RAZ_WIN_WHITE : RAZ_WIN_DRAW);
                    }
                }
                np |= p2 << (8 * i);
                ns |= s2 << (8 * i);
            });
            npw[q] = np;
            nsw[q] = ns;
        });
        sl::sfor<16>([&](auto r_) __attribute__((always_inline)) {
            constexpr int r = decltype(r_)::value;
            const uint32_t hp = npw[r >> 1] >> (16 * (r & 1)), hs = nsw[r >> 1] >> (16 * (r & 1));
            player2[at + (size_t)r * 64] = make_uchar2((unsigned char)(hp & 0xff), (unsigned char)((hp >> 8) & 0xff));
            status2[at + (size_t)r * 64] = make_uchar2((unsigned char)(hs & 0xff), (unsigned char)((hs >> 8) & 0xff));
        });
    }
}

// ReversiEnv.step, HYBRID: the move itself board by board with the reference-shaped ray arithmetic (bbv_calc_flip: exact for every input,
// 111 instructions per board), the legal moves after it - two find_correct_moves per board in the board-per-lane kernel's worst case,
// 134 each - bit-sliced for the lane's 32 boards at once (raz_sweep_sliced.h legal_after_move: 2 transposes in, 1 out, 41 for both
// mobilities): 262 vector instructions per board against 344.  A wave takes a superblock of 2048 boards, a lane 32 of them (rows r = 0..15
// of 64 lanes x 2 boards, as k_step_sliced).  What makes it fit 256 registers without scratch, so that two waves per SIMD overlap:
//   * the boards after the move go to memory as soon as a row is made and stay in registers only as the sliced phase's input (a row's 8
//     registers of loaded boards become 8 of own' / enemy' words);
//   * the player / status / action bytes come in as six 16-byte loads per lane, are handed round through LDS, and what the epilogue needs
//     of them waits in LDS (one word per row) while the sliced phase has the registers; the winners by disc count are 2 bits per board;
//   * every stream is addressed by a scalar base per 4 KB + ONE vector register (the lane's offset);
//   * values the optimiser would keep alive across the phases (store-to-load forwarding through LDS, the disc counts sunk to their rare
//     use, sixteen partial masks instead of one, the lane's number) are pinned where they belong (RAZ_SL_PIN, lane_again).
// All of a superblock's loads are issued before the first row is worked on.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_step_hybrid(
    ulonglong2* __restrict__ black2, ulonglong2* __restrict__ white2, uchar2* __restrict__ player2, uchar2* __restrict__ status2,
    ulonglong2* __restrict__ legal2, const uchar2* __restrict__ action2, size_t nsb) {
    for (size_t sb = blockIdx.x; sb < nsb; sb += gridDim.x) {
        const unsigned lane = threadIdx.x;
        // ALL of the superblock's loads are issued before anything is worked on (the boards' registers are free until the rows turn them
        // into own' / enemy' words): the superblock's arrays by a scalar base per 4 rows = 4 KB of a stream, the reach of an instruction's
        // immediate offset, + the lane's 32-bit offset - one address register for all 16 rows of all the streams
        uint4 p0, p1, s0, s1, a0, a1;
        ulonglong2 bq[16], wq[16];
        {
            const uint4 *const pp = (const uint4*)sl::uniform_ptr(player2 + sb * 1024), *const sp = (const uint4*)sl::uniform_ptr(status2 + sb * 1024),
                        *const ap = (const uint4*)sl::uniform_ptr(action2 + sb * 1024);
            p0 = pp[lane]; p1 = pp[64u + lane]; s0 = sp[lane]; s1 = sp[64u + lane]; a0 = ap[lane]; a1 = ap[64u + lane];
            sl::sfor<4>([&](auto g_) __attribute__((always_inline)) {
                constexpr int g = decltype(g_)::value;
                const ulonglong2 *const bg = sl::uniform_ptr(black2 + sb * 1024 + g * 256), *const wg = sl::uniform_ptr(white2 + sb * 1024 + g * 256);
                sl::sfor<4>([&](auto i_) __attribute__((always_inline)) {
                    constexpr int i = decltype(i_)::value;
                    bq[4 * g + i] = bg[i * 64u + lane]; wq[4 * g + i] = wg[i * 64u + lane];
                });
            });
        }
        ulonglong2 *bl[4], *wh[4], *lg[4];
        sl::sfor<4>([&](auto g_) __attribute__((always_inline)) {
            constexpr int g = decltype(g_)::value;
            bl[g] = sl::uniform_ptr(black2 + sb * 1024 + g * 256); wh[g] = sl::uniform_ptr(white2 + sb * 1024 + g * 256);
            lg[g] = sl::uniform_ptr(legal2 + sb * 1024 + g * 256);
        });
        uchar2 *const pl = sl::uniform_ptr(player2 + sb * 1024), *const st2 = sl::uniform_ptr(status2 + sb * 1024);
        uint32_t o[64], e[64], L[64];
        // what the epilogue needs of every board besides the masks - its player byte and the status it leaves with if nothing moved (a game
        // already over, a resignation, an illegal action) - waits in LDS ([row][lane]: conflict-free) while the sliced phase has the
        // registers; the winner by disc count, should the game end with this move, is two bits per board of winsw
        __shared__ uint32_t stash[16][64];
        uint32_t winsw[2] = {0u, 0u};
        // the superblock's player / status / action bytes: six 16-byte loads per lane instead of 48 two-byte ones, handed round through LDS
        __shared__ __attribute__((aligned(16))) uint8_t psa[3][2048];
        uint32_t moved = 0u;
        // rows are worked on one after the other as they arrive (the scheduler must not interleave the 32 boards' flips: their
        // temporaries would not fit); a row's 8 registers of boards turn into 8 of own' / enemy' words
        RAZ_SL_PHASE();
        __syncthreads();   // (every lane has read the bytes of the superblock before; a one-wave workgroup: no instruction)
        *(uint4*)&psa[0][lane * 16] = p0; *(uint4*)&psa[0][1024 + lane * 16] = p1;
        *(uint4*)&psa[1][lane * 16] = s0; *(uint4*)&psa[1][1024 + lane * 16] = s1;
        *(uint4*)&psa[2][lane * 16] = a0; *(uint4*)&psa[2][1024 + lane * 16] = a1;
        __syncthreads();
        sl::sfor<16>([&](auto r_) __attribute__((always_inline)) {
            constexpr int r = decltype(r_)::value;
            const uchar2 pv = *(const uchar2*)&psa[0][(r * 64 + lane) * 2], sv = *(const uchar2*)&psa[1][(r * 64 + lane) * 2],
                         av = *(const uchar2*)&psa[2][(r * 64 + lane) * 2];
            RAZ_SL_PHASE();
            const ulonglong2 bv = bq[r], wv = wq[r];
            ulonglong2 nbv, nwv;
            uint32_t keep = (uint32_t)pv.x | ((uint32_t)pv.y << 8);
            sl::sfor<2>([&](auto j_) __attribute__((always_inline)) {
                constexpr int j = decltype(j_)::value, k = 2 * r + j;
                const raz_bb b = j ? bv.y : bv.x, w = j ? wv.y : wv.x;
                const uint32_t p = j ? pv.y : pv.x, st = j ? sv.y : sv.x, a = j ? av.y : av.x;
                const bool blk = p == RAZ_PLAYER_BLACK;
                const raz_bb own = blk ? b : w, enemy = blk ? w : b;
                raz_bb fl = bbv_calc_flip((int)(a & 63u), own, enemy);
                fl = (st == 0u && a < 64u) ? fl : 0ULL;   // a finished game, a resignation, an action outside the board: nothing moves
                const bool mv = fl != 0ULL;
                const raz_bb own2 = mv ? ((own ^ fl) | (1ULL << (a & 63u))) : own, enemy2 = enemy ^ fl;
                const raz_bb nb = blk ? own2 : enemy2, nw = blk ? enemy2 : own2;
                if (j) { nbv.y = nb; nwv.y = nw; } else { nbv.x = nb; nwv.x = nw; }
                o[k] = (uint32_t)own2; o[32 + k] = (uint32_t)(own2 >> 32);
                e[k] = (uint32_t)enemy2; e[32 + k] = (uint32_t)(enemy2 >> 32);
                moved |= (mv ? 1u : 0u) << k;
                const int cb = bb_popcount(nb), cw = bb_popcount(nw);
                winsw[k >> 4] |= (uint32_t)(cb > cw ? RAZ_WIN_BLACK : (cb < cw ? RAZ_WIN_WHITE : RAZ_WIN_DRAW)) << (2 * (k & 15));
                const uint32_t other_wins = blk ? RAZ_WIN_WHITE : RAZ_WIN_BLACK;
                const uint32_t unmoved = st != 0u ? st : (other_wins | (a == RAZ_ACTION_RESIGN ? RAZ_STATUS_RESIGNED : RAZ_STATUS_ILLEGAL));
                keep |= unmoved << (16 + 8 * j);
            });
            bl[r >> 2][(r & 3) * 64u + lane] = nbv;
            wh[r >> 2][(r & 3) * 64u + lane] = nwv;
            stash[r][lane] = keep;
            RAZ_SL_PIN(winsw[r >> 3]);   // (counted HERE, while the boards are in registers - not sunk to the epilogue's rare use)
            RAZ_SL_PIN(moved);           // (ONE word: not sixteen partial ones kept for the epilogue's bit tests)
            RAZ_SL_PHASE();
        });
        RAZ_SL_PHASE();
        sl::transpose_boards(o);
        sl::transpose_boards(e);
        uint32_t nz1, nz2;
        sl::legal_after_move(o, e, L, moved, nz1, nz2);
        sl::transpose_boards(L);
        RAZ_SL_PHASE();
        const unsigned lane_ = sl::lane_again();
        sl::sfor<16>([&](auto r_) __attribute__((always_inline)) {
            constexpr int r = decltype(r_)::value;
            ulonglong2 v;
            v.x = ((unsigned long long)L[32 + 2 * r] << 32) | L[2 * r];
            v.y = ((unsigned long long)L[32 + 2 * r + 1] << 32) | L[2 * r + 1];
            lg[r >> 2][(r & 3) * 64u + lane_] = v;
        });
        RAZ_SL_PHASE();
        // player / status of every board (bb_env_step's cases, without branches): a board that moved hands the turn over where the
        // opponent can move (nz1), ends with the disc count's winner where neither side can (nz2 clear), and goes on with the same
        // player otherwise; a board that did not move leaves with the status phase 1 put by
        asm volatile("" ::: "memory");   // (the stash is READ here: the words must not be carried over the sliced phase in registers)
        const uint32_t turn = moved & nz1, ends = moved & ~nz1 & ~nz2;
        sl::sfor<16>([&](auto r_) __attribute__((always_inline)) {
            constexpr int r = decltype(r_)::value;
            const uint32_t keep = stash[r][lane_];
            uint32_t hp = 0u, hs = 0u;
            sl::sfor<2>([&](auto j_) __attribute__((always_inline)) {
                constexpr int j = decltype(j_)::value, k = 2 * r + j;
                const uint32_t p = (keep >> (8 * j)) & 0xffu, unmoved = (keep >> (16 + 8 * j)) & 0xffu;
                const uint32_t wins = (winsw[k >> 4] >> (2 * (k & 15))) & 3u;
                const uint32_t p2 = (turn >> k) & 1u ? (3u - p) & 0xffu : p;
                const uint32_t s2 = (moved >> k) & 1u ? ((ends >> k) & 1u ? wins : 0u) : unmoved;
                hp |= p2 << (8 * j);
                hs |= s2 << (8 * j);
            });
            pl[r * 64u + lane_] = make_uchar2((unsigned char)(hp & 0xff), (unsigned char)(hp >> 8));
            st2[r * 64u + lane_] = make_uchar2((unsigned char)(hs & 0xff), (unsigned char)(hs >> 8));
        });
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

// Which batches run their whole superblocks (2048 boards = one wave's work) on the bit-sliced kernels - measured on an MI355X
// (profiles/r6/sweep_bit_sliced_vs_board_per_lane_ab.jsonl, profiles/r6/sweep_step_hybrid_vs_board_per_lane_ab.jsonl):
//   find_correct_moves: 66 instead of 134 vector instructions per board, two waves per SIMD.  2^26 boards: 0.279-0.291 ms against
//     0.315-0.325 ms (5.8 against 5.0 TB/s); 2^24 boards: 0.075-0.078 against 0.070-0.072 ms - a superblock per wave leaves too few
//     waves to overlap one wave's loads with another's arithmetic.  Default: from 2^25 boards on.
//   ReversiEnv.step, the board-per-lane kernel: 344 instructions per board, at its instruction-issue floor (0.149-0.153 ms at 2^24 boards,
//     0.61-0.64 ms at 2^26).
//     k_step_hybrid (the move per board, the legal moves after it sliced): 262 per board, 256 registers + 10 KB LDS, no scratch, two waves
//     per SIMD: 0.577-0.601 ms at 2^26 boards (0.63-0.655 of 8 TB/s against 0.595-0.615), 0.169-0.175 ms at 2^24 (four rounds of 2048
//     waves that start together do not overlap as thousands of small waves do).  Default: from 2^26 boards on.
//     k_step_sliced (everything sliced): 247 per board but 256 + 176 registers = ONE wave per SIMD, whose loads, arithmetic and stores
//     do not overlap: 0.200 ms at 2^24, 0.70 ms at 2^26; compiled for two waves (234 registers in scratch) 0.32 ms.  Kept, parity-tested.
// RAZ_SWEEP_SLICED_STEP: unset = as above; 0 = the board-per-lane kernel always; 1 = k_step_sliced, 2 = k_step_hybrid from 2^21 boards on.
// RAZ_SWEEP_SLICED_MIN=<boards> overrides every threshold (tests: 2048 runs every whole superblock on the sliced kernels).
inline size_t env_boards(const char* name, size_t otherwise) {
    const char* s = getenv(name);
    const size_t v = s && *s ? (size_t)strtoull(s, nullptr, 10) : otherwise;
    return v < 2048 ? 2048 : v;
}
inline size_t sliced_min_boards() {   // find_correct_moves
    static const size_t v = env_boards("RAZ_SWEEP_SLICED_MIN", (size_t)1 << 25);
    return v;
}
inline int step_form() {   // -1 = unset
    static const int v = [] {
        const char* on = getenv("RAZ_SWEEP_SLICED_STEP");
        return on && *on >= '0' && *on <= '2' ? *on - '0' : -1;
    }();
    return v;
}
inline size_t sliced_step_min_boards() {
    static const size_t v = env_boards("RAZ_SWEEP_SLICED_MIN", step_form() < 0 ? (size_t)1 << 26 : (size_t)1 << 21);
    return v;
}
// k_step_hybrid's grid: a wave per superblock (RAZ_SWEEP_HYBRID_WAVES=<n>: n waves walk the superblocks - tests, so that a wave takes
// several.  Measured: 2048 resident waves walking all the superblocks, each issuing its next one's loads before the stores of the one
// at hand, are SLOWER - 0.203 against 0.170 ms at 2^24 boards, 0.70 against 0.58 ms at 2^26 - and so is starting the two waves of a
// SIMD half a wave's life apart)
inline size_t hybrid_waves() {
    static const size_t v = [] {
        const char* w = getenv("RAZ_SWEEP_HYBRID_WAVES");
        const size_t v = w && *w ? (size_t)strtoull(w, nullptr, 10) : (size_t)kMaxGrid;
        return v ? v : 1;
    }();
    return v;
}

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
    // large batches: whole superblocks of 2048 boards on the bit-sliced kernel, the rest on the board-per-lane kernel
    const size_t nsb = n >= sliced_min_boards() ? n / 2048 : 0, done = nsb * 2048;
    if (nsb)
        hipLaunchKernelGGL(k_legal_moves_sliced, dim3((unsigned)(nsb > (size_t)kMaxGrid ? (size_t)kMaxGrid : nsb)), dim3(64), 0, (hipStream_t)stream,
                           (const ulonglong2*)own, (const ulonglong2*)enemy, (ulonglong2*)legal, nsb);
    if (n > done)
        hipLaunchKernelGGL(k_legal_moves, dim3(grid_for((n - done) / 4 + 1)), dim3(kBlock), 0, (hipStream_t)stream,
                           (const ulonglong2*)(own + done), (const ulonglong2*)(enemy + done), (ulonglong2*)(legal + done),
                           (const raz_bb*)(own + done), (const raz_bb*)(enemy + done), (raz_bb*)(legal + done), n - done);
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
    // whole superblocks on the sliced form in force (k_step_hybrid reads the byte arrays 16 bytes at a time: other alignments stay on the
    // board-per-lane kernel), the rest a board per lane
    int form = step_form() < 0 ? 2 : step_form();
    if (form == 2 && !(aligned16(player) && aligned16(status) && aligned16(action))) form = 0;
    const size_t nsb = form && n >= sliced_step_min_boards() ? n / 2048 : 0, done = nsb * 2048;
    if (nsb && form == 2)
        hipLaunchKernelGGL(k_step_hybrid, dim3((unsigned)(nsb > hybrid_waves() ? hybrid_waves() : nsb)), dim3(64), 0, (hipStream_t)stream,
                           (ulonglong2*)black, (ulonglong2*)white, (uchar2*)player, (uchar2*)status, (ulonglong2*)legal, (const uchar2*)action, nsb);
    else if (nsb)
        hipLaunchKernelGGL(k_step_sliced, dim3((unsigned)(nsb > (size_t)kMaxGrid ? (size_t)kMaxGrid : nsb)), dim3(64), 0, (hipStream_t)stream,
                           (ulonglong2*)black, (ulonglong2*)white, (uchar2*)player, (uchar2*)status, (ulonglong2*)legal, (const uchar2*)action, nsb);
    if (n > done)
        hipLaunchKernelGGL(k_step, dim3(grid_for((n - done) / 4 + 1)), dim3(kBlock), 0, (hipStream_t)stream,
                           (raz_bb*)black + done, (raz_bb*)white + done, player + done, status + done, (raz_bb*)legal + done, action + done, n - done);
    return raz_check_launch("raz_step_batch");
}

extern "C" int raz_sweep_forms(size_t n, int* legal_moves_form, int* step_form_out) {
    RAZ_REQUIRE(legal_moves_form && step_form_out, "raz_sweep_forms: NULL result");
    *legal_moves_form = n >= sliced_min_boards() && n >= 2048 ? 1 : 0;
    const int form = step_form() < 0 ? 2 : step_form();
    *step_form_out = form && n >= sliced_step_min_boards() && n >= 2048 ? form : 0;
    return RAZ_OK;
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
