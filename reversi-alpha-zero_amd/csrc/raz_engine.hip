// raz_engine.hip — the batched MCTS self-play engine: ONE WAVEFRONT PER GAME, LANE = BOARD SQUARE.
//
// What the reference does per game in Python (agent/player.py) and what happens here:
//   * the three 64-vectors N / W / P of a position (player.py:62-69) are stored for its LEGAL moves only, in one
//     compact variable-size node in HBM (raz_engine.h); a wave loads them in one round trip, lane r holding the
//     r-th legal move;
//   * PUCT (select_action_q_and_u, :395-428) is per-lane f32/f64 arithmetic in the reference's
//     exact dtype order + wave reductions (exact integer sum of N, numpy's pairwise f32 sum of p,
//     first-maximum argmax);
//   * the board ops (find_correct_moves / calc_flip / step) are wave-uniform 64-bit integer code;
//   * the transposition "dict" is a per-game open-addressing table probed 16 slots per request;
//   * one kernel launch = for every game: back up the previous leaf (:264-281), run the per-move
//     controller when a search completes (action_with_evaluation :82-134, start_game
//     worker/self_play.py:155-162), then descend to the next leaf (search_my_move :217-262) and
//     emit it for the cross-game net batch.  Terminal leaves need no net, so a game may complete
//     several simulations inside one launch.
//
// Bit-exactness contract: every floating-point operation below is written in the order the
// reference performs it under numpy 2 and the file is compiled with
// -ffp-contract=off; tests/test_engine_gpu.py compares full games with the CPU oracle, which is
// itself pinned to the unmodified reference.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include <new>
#include <mutex>
#include <vector>
#include "raz_bitboard.h"
#include "raz_detmath.h"
#include "raz_engine.h"
#include "raz_internal.h"

namespace {

constexpr int kInnerMax = 2;  // max simulations completed per game per launch (terminal leaves need no net);
                              // larger values make the few games with runs of terminal leaves stragglers
                              // that set the launch's duration (8: 58.6M sims/s, 2: 62.2M on the bench config)

// ------------------------------------------------------------------ wave helpers
// Lanes of the game's wave communicate through HBM (lane 0 writes game state, all lanes read it).
// Each lane is a separate thread to the compiler, so such hand-offs need an acquire/release point
// or an earlier load may be forwarded past another lane's store.  All communicating lanes belong
// to ONE wavefront, whose vector-memory operations execute in program order, so a wavefront-scope
// fence is sufficient: it constrains the compiler and emits no instruction (LLVM AMDGPU memory
// model, gfx942 table: "fence acq_rel - wavefront: none") — in particular no s_waitcnt vmcnt(0)
// that would stall the wave until its stores are acknowledged.
__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

// Values every lane holds identically (game state, keys, node indices) are moved to SGPRs so the
// 64-bit board arithmetic and the address math run on the scalar unit.
__device__ __forceinline__ uint32_t uni(uint32_t x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ int uni(int x) { return (int)__builtin_amdgcn_readfirstlane((uint32_t)x); }
__device__ __forceinline__ raz_bb uni(raz_bb x) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)x);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(x >> 32));
    return ((raz_bb)hi << 32) | lo;
}
__device__ __forceinline__ uint32_t lane_u32(uint32_t v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ double lane_f64(double v, int l) {
    const uint64_t b = raz_f64_to_bits(v);
    const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)b, l), hi = __builtin_amdgcn_readlane((uint32_t)(b >> 32), l);
    return raz_bits_to_f64(((uint64_t)hi << 32) | lo);
}

// DPP lane permutations inside a row of 16: xor 1, xor 2 (quad_perm), then half-mirror and mirror,
// which act as xor 4 / xor 8 once the smaller groups already agree.  VALU-rate, no LDS crossbar.
template <int CTRL>
__device__ __forceinline__ uint32_t dpp32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}
template <int CTRL>
__device__ __forceinline__ double dpp64(double v) {
    const uint64_t b = raz_f64_to_bits(v);
    const uint32_t lo = dpp32<CTRL>((uint32_t)b), hi = dpp32<CTRL>((uint32_t)(b >> 32));
    return raz_bits_to_f64(((uint64_t)hi << 32) | lo);
}
#define RAZ_DPP_XOR1 0xB1
#define RAZ_DPP_XOR2 0x4E
#define RAZ_DPP_HALF_MIRROR 0x141
#define RAZ_DPP_MIRROR 0x140

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {  // exact, order-free
    v += dpp32<RAZ_DPP_XOR1>(v);
    v += dpp32<RAZ_DPP_XOR2>(v);
    v += dpp32<RAZ_DPP_HALF_MIRROR>(v);
    v += dpp32<RAZ_DPP_MIRROR>(v);
    return lane_u32(v, 0) + lane_u32(v, 16) + lane_u32(v, 32) + lane_u32(v, 48);
}

// np.sum over float32[64] in numpy's pairwise order: 8 running partials r[j] += a[8i+j], then
// ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)).  Once per node (at expansion).  The vector goes through
// 256 B of LDS so that lane j can read its column a[j], a[8+j], ... with eight independent loads.
__device__ __forceinline__ float wave_np_sum_f32(float a, int lane, float* lds64) {
    lds64[lane] = a;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();  // single-wave workgroup: orders the LDS writes before the reads
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const int j = lane & 7;
    float t = lds64[j];
#pragma unroll
    for (int i = 1; i < 8; ++i) t = t + lds64[j + 8 * i];
    t = t + raz_bits_to_f32(dpp32<RAZ_DPP_XOR1>(__float_as_uint(t)));
    t = t + raz_bits_to_f32(dpp32<RAZ_DPP_XOR2>(__float_as_uint(t)));
    t = t + raz_bits_to_f32(dpp32<RAZ_DPP_HALF_MIRROR>(__float_as_uint(t)));
    return raz_bits_to_f32(lane_u32(__float_as_uint(t), 0));
}

// argmax with numpy's first-maximum rule; the result is wave-uniform.
#define RAZ_ARGMAX_STEP(CTRL)                                   \
    {                                                           \
        const double ov = dpp64<CTRL>(v);                       \
        const int oi = (int)dpp32<CTRL>((uint32_t)idx);         \
        if (ov > v || (ov == v && oi < idx)) {                  \
            v = ov;                                             \
            idx = oi;                                           \
        }                                                       \
    }
__device__ __forceinline__ int wave_argmax_f64(double v, int lane) {
    int idx = lane;
    RAZ_ARGMAX_STEP(RAZ_DPP_XOR1)
    RAZ_ARGMAX_STEP(RAZ_DPP_XOR2)
    RAZ_ARGMAX_STEP(RAZ_DPP_HALF_MIRROR)
    RAZ_ARGMAX_STEP(RAZ_DPP_MIRROR)
    double bv = lane_f64(v, 0);
    int bi = (int)lane_u32((uint32_t)idx, 0);
#pragma unroll
    for (int r = 16; r < 64; r += 16) {
        const double ov = lane_f64(v, r);
        const int oi = (int)lane_u32((uint32_t)idx, r);
        if (ov > bv || (ov == bv && oi < bi)) {
            bv = ov;
            bi = oi;
        }
    }
    return bi;
}
// argmax of NON-NEGATIVE finite doubles (the PUCT scores: (+-Q + U + 1000) * legal >= 0): their IEEE
// bit patterns order like unsigned integers, so the maximum is found with two u32 max-reductions
// (high words, then low words among the lanes that hold the top high word) and the first maximum
// with a ballot — a third of the instructions of the compare-and-select f64 reduction above.
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    v = max(v, dpp32<RAZ_DPP_XOR1>(v));
    v = max(v, dpp32<RAZ_DPP_XOR2>(v));
    v = max(v, dpp32<RAZ_DPP_HALF_MIRROR>(v));
    v = max(v, dpp32<RAZ_DPP_MIRROR>(v));
    return max(max(lane_u32(v, 0), lane_u32(v, 16)), max(lane_u32(v, 32), lane_u32(v, 48)));
}
__device__ __forceinline__ int wave_argmax_nonneg_f64(double v) {
    const uint64_t b = raz_f64_to_bits(v);
    const uint32_t hi = (uint32_t)(b >> 32), lo = (uint32_t)b;
    const uint32_t mh = wave_max_u32(hi);
    const bool top = hi == mh;
    const uint32_t ml = wave_max_u32(top ? lo : 0u);
    return __ffsll((long long)__ballot(top && lo == ml)) - 1;
}
__device__ __forceinline__ double wave_max_f64(double v) {
    double o;
    o = dpp64<RAZ_DPP_XOR1>(v); v = o > v ? o : v;
    o = dpp64<RAZ_DPP_XOR2>(v); v = o > v ? o : v;
    o = dpp64<RAZ_DPP_HALF_MIRROR>(v); v = o > v ? o : v;
    o = dpp64<RAZ_DPP_MIRROR>(v); v = o > v ? o : v;
    double b = lane_f64(v, 0);
#pragma unroll
    for (int r = 16; r < 64; r += 16) {
        const double x = lane_f64(v, r);
        b = x > b ? x : b;
    }
    return b;
}

// Optional phase profile (cfg.reserved & 1): shader-clock ticks (s_memtime) accumulated per game:
// [0] backup, [1] controller, [2] select, [3] root-noise sampling, [4] first-arrival probes,
// [5] launches in which the game did work, [6] node-vector load waits in select, [7] expansion part of backup.
#define RAZ_PROF_ON(E) ((E).cfg.reserved & 1u)
__device__ __forceinline__ unsigned long long prof_now() { return __builtin_amdgcn_s_memtime(); }
__device__ __forceinline__ void prof_add(const raz_engine_dev& E, uint32_t g, int k, unsigned long long t0, int lane) {
    if (RAZ_PROF_ON(E) && lane == 0) E.prof[(size_t)g * 8 + k] += prof_now() - t0;
}

// ------------------------------------------------------------------ tree storage
__device__ __forceinline__ uint32_t key_hash(raz_bb b, raz_bb w, uint32_t tagkey) {
    unsigned long long x = b * 0x9E3779B97F4A7C15ULL ^ (w + tagkey) * 0xC2B2AE3D27D4EB4FULL;
    x ^= x >> 29;
    x *= 0xBF58476D1CE4E5B9ULL;
    x ^= x >> 32;
    return (uint32_t)x;
}

struct Found {        // all fields wave-uniform
    bool found;
    uint32_t node;    // the node's link, valid when found
    uint32_t slot;    // first empty slot when !found; 0xffffffff = table full
};

__device__ Found table_find(const raz_engine_dev& E, uint32_t g, raz_bb b, raz_bb w, uint32_t tagkey,
                            int lane) {
    const raz_slot* tab = E.table + (size_t)g * E.H;
    const uint32_t mask = E.H - 1;
    const uint32_t h = key_hash(b, w, tagkey);
    Found f;
    f.found = false;
    f.node = 0;
    f.slot = 0xffffffffu;
    for (uint32_t r = 0; r < E.H; r += RAZ_PROBE) {
        const uint32_t si = (h + r + (uint32_t)(lane & (RAZ_PROBE - 1))) & mask;
        const raz_slot* s = tab + si;
        const raz_bb sb = s->black, sw = s->white;
        const uint32_t it = s->idx_tag, lk = s->link;
        const bool used = (it & RAZ_SLOT_USED) != 0;
        const bool match = used && sb == b && sw == w && (it & RAZ_SLOT_KEYMASK) == tagkey;
        const unsigned long long mm = __ballot(match) & 0xffffULL;
        const unsigned long long em = __ballot(!used) & 0xffffULL;
        if (mm) {
            const int jj = __ffsll((long long)mm) - 1;
            f.found = true;
            f.node = lane_u32(lk, jj);
            return f;
        }
        if (em) {
            f.slot = (h + r + (uint32_t)(__ffsll((long long)em) - 1)) & mask;
            return f;
        }
    }
    return f;
}

// ---- compact nodes (raz_engine.h): a LINK = (byte offset / 8) << 6 | L names a node and tells its array length
__device__ __forceinline__ int link_L(uint32_t link) { return (int)(link & 63u); }
__device__ __forceinline__ uint32_t link_make(uint32_t off8, int L) { return (off8 << 6) | (uint32_t)L; }
__device__ __forceinline__ uint32_t node_units(int L) { return (uint32_t)(RAZ_NODE_HDR_BYTES + RAZ_NODE_ENTRY_BYTES * L + 7) >> 3; }
__device__ __forceinline__ unsigned char* node_ptr(const raz_engine_dev& E, uint32_t g, uint32_t link) {
    return E.nodes + (size_t)g * E.pool_bytes + (size_t)(link >> 6) * 8;
}
__device__ __forceinline__ raz_node_hdr* node_hdr(unsigned char* p) { return (raz_node_hdr*)p; }
__device__ __forceinline__ double* node_W(unsigned char* p, int L) { (void)L; return (double*)(p + RAZ_NODE_HDR_BYTES); }
__device__ __forceinline__ uint32_t* node_N(unsigned char* p, int L) { return (uint32_t*)(p + RAZ_NODE_HDR_BYTES + 8 * L); }
__device__ __forceinline__ float* node_P(unsigned char* p, int L) { return (float*)(p + RAZ_NODE_HDR_BYTES + 12 * L); }
__device__ __forceinline__ uint32_t* node_child(unsigned char* p, int L) { return (uint32_t*)(p + RAZ_NODE_HDR_BYTES + 16 * L); }
// rank of square `sq` among the set bits of `legal` (= index of its entry in the node's arrays)
__device__ __forceinline__ int rank_of(raz_bb legal, int sq) { return __popcll(legal & ((1ULL << sq) - 1ULL)); }
// the square of the r-th (0-based) legal move; r < popcount(legal).  Wave-uniform arguments: one ballot.
__device__ __forceinline__ int square_of_rank(raz_bb legal, int r, int lane) {
    return __ffsll((long long)__ballot(((legal >> lane) & 1ULL) && rank_of(legal, lane) == r)) - 1;
}
// A node's vectors in SQUARE space (lane = board square, zero off the legal moves): what the per-move controller and the
// records want (player.py:62-69 indexes var_n / var_w by action).
__device__ __forceinline__ void node_read_squares(unsigned char* p, int L, raz_bb legal, int lane, double& Wi, uint32_t& Ni) {
    const bool on = (legal >> lane) & 1ULL;
    const int rk = rank_of(legal, lane);
    Wi = on ? node_W(p, L)[rk] : 0.0;
    Ni = on ? node_N(p, L)[rk] : 0u;
}

// ------------------------------------------------------------------ the game's control block in registers
// raz_game (raz_engine.h) is 64 dwords.  k_tree loads it with ONE coalesced request (lane i = dword
// i) together with the in-flight path and the net's answer, keeps it in a single VGPR for the whole
// launch — a field read is one v_readlane, a field write one v_writelane, no memory round trip, no
// fence — and stores it back once at the end.  (The first version kept every field in its own
// [B] array: ~40 dependent loads per simulation at 120-500 cycles each were 60 % of the kernel.)
#define GW(field) ((int)(offsetof(raz_game, field) / 4))
struct Regs {
    uint32_t cw;                    // lane i = dword i of raz_game
    uint32_t pnode, pmirror, pact;  // path of the simulation in flight: lane d = level d
    float pol_raw, val;             // the net's policy row (lane = square of the TRANSFORMED board) and value
    uint32_t nn;                    // this launch produced a leaf for the net
    uint32_t path_dirty;
};
__device__ __forceinline__ uint32_t G32(const Regs& R, int i) { return (uint32_t)__builtin_amdgcn_readlane(R.cw, i); }
__device__ __forceinline__ raz_bb G64(const Regs& R, int i) {  // (the builtin returns int: widen through uint32_t)
    return (raz_bb)(uint32_t)__builtin_amdgcn_readlane(R.cw, i) | ((raz_bb)(uint32_t)__builtin_amdgcn_readlane(R.cw, i + 1) << 32);
}
// v_writelane_b32: this clang has no __builtin for it, so the LLVM intrinsic is bound by name; the
// compiler then sees a VALU instruction and inserts the gfx950 wait states around it (VALU-written
// SGPR -> VALU read: 2; VALU-written VGPR -> v_readlane: 1), which inline asm would hide from it.
#ifndef RAZ_WAVE_EMU   // (tests/native/wave_emu steps these kernels on the host and supplies its own)
extern "C" __device__ uint32_t raz_llvm_writelane(uint32_t src, uint32_t lane, uint32_t old) __asm("llvm.amdgcn.writelane.i32");
#endif
template <int I>
__device__ __forceinline__ uint32_t writelane_c(uint32_t old, uint32_t v) {
    static_assert(I >= 0 && I < 64, "lane");
    return raz_llvm_writelane(__builtin_amdgcn_readfirstlane(v), (uint32_t)I, old);
}
__device__ __forceinline__ uint32_t writelane_r(uint32_t old, uint32_t v, int at, int lane) {
    (void)lane;
    return raz_llvm_writelane(__builtin_amdgcn_readfirstlane(v), __builtin_amdgcn_readfirstlane((uint32_t)at), old);
}
template <int I>
__device__ __forceinline__ void set32(Regs& R, uint32_t v) { R.cw = writelane_c<I>(R.cw, v); }
template <int I>
__device__ __forceinline__ void set64(Regs& R, raz_bb v) {
    R.cw = writelane_c<I>(R.cw, (uint32_t)v);
    R.cw = writelane_c<I + 1>(R.cw, (uint32_t)(v >> 32));
}
#define S32(R, I, v) set32<(I)>((R), (uint32_t)(v))
#define S64(R, I, v) set64<(I)>((R), (raz_bb)(v))
#define ADD64(R, I, d) S64(R, I, G64(R, I) + (raz_bb)(d))
__device__ __forceinline__ void flag_error(Regs& R, uint32_t f) { S32(R, GW(error), G32(R, GW(error)) | f); }

// Reserve room for a node with L legal moves at the end of the game's pool: `used` (8-byte units) and `count` are the
// caller's running copies of pool_used / node_count.  Returns the new node's link, or RAZ_NO_NODE when the pool's bytes
// or its node count are exhausted (the caller flags the error).
__device__ __forceinline__ uint32_t pool_take(const raz_engine_dev& E, int L, uint32_t& used, uint32_t& count) {
    const uint32_t units = node_units(L);
    if (count >= E.C || (unsigned long long)(used + units) * 8ULL > E.pool_bytes) return RAZ_NO_NODE;
    const uint32_t link = link_make(used, L);
    used += units;
    ++count;
    return link;
}

// Write node `link` (just taken from the pool as the game's node number `index`) for key (b, w, tag) into the EMPTY table
// slot `slot`: statistics zero, prior `prior_sq` given in SQUARE space (lane = square; only the legal lanes store).
// Pure stores: nothing is read back.
__device__ __forceinline__ void node_init(const raz_engine_dev& E, uint32_t g, uint32_t link, uint32_t index, uint32_t slot,
                                          raz_bb b, raz_bb w, uint32_t tag, raz_bb legal, uint32_t mirror,
                                          float prior_sq, int lane) {
    unsigned char* p = node_ptr(E, g, link);
    const int L = link_L(link);
    if ((legal >> lane) & 1ULL) {
        const int rk = rank_of(legal, lane);
        node_W(p, L)[rk] = 0.0;
        node_N(p, L)[rk] = 0u;
        node_P(p, L)[rk] = prior_sq;
        node_child(p, L)[rk] = 0u;
    }
    if (lane == 0) {
        raz_node_hdr h;
        h.black = b;
        h.white = w;
        h.legal = legal;
        h.tag = tag;
        h.mirror = mirror;
        h.index = index;
        h.gc_index = 0;
        *node_hdr(p) = h;
        E.node_dir[(size_t)g * E.C + index] = link;
        raz_slot* s = E.table + (size_t)g * E.H + slot;
        s->black = b;
        s->white = w;
        s->link = link;
        s->idx_tag = RAZ_SLOT_USED | (tag & RAZ_SLOT_KEYMASK);
    }
}

// Allocate a zeroed node for key (b, w, np, owner) in the EMPTY table slot `slot`.
// Returns RAZ_NO_NODE after flagging an error when out of space.
__device__ uint32_t node_create_at(const raz_engine_dev& E, Regs& R, uint32_t g, uint32_t slot, raz_bb b, raz_bb w,
                                   uint32_t np, uint32_t owner, raz_bb legal, int lane) {
    uint32_t used = G32(R, GW(pool_used)), count = G32(R, GW(node_count));
    const uint32_t link = slot == 0xffffffffu ? RAZ_NO_NODE : pool_take(E, __popcll(legal), used, count);
    if (link == RAZ_NO_NODE) {
        flag_error(R, (slot == 0xffffffffu) ? RAZ_ERR_TABLE_FULL : RAZ_ERR_POOL_FULL);
        return RAZ_NO_NODE;
    }
    node_init(E, g, link, count - 1, slot, b, w, np | (owner << 2), legal, RAZ_NO_NODE, 0.0f, lane);
    S32(R, GW(pool_used), used);
    S32(R, GW(node_count), count);
    wave_sync();
    return link;
}

// defaultdict access: find the node of (b, w, np) for `owner`, creating a zeroed one if absent.
__device__ uint32_t node_get(const raz_engine_dev& E, Regs& R, uint32_t g, raz_bb b, raz_bb w, uint32_t np,
                             uint32_t owner, raz_bb legal, int lane) {
    const Found f = table_find(E, g, b, w, np | (owner << 2), lane);
    if (f.found) return f.node;
    return node_create_at(E, R, g, f.slot, b, w, np, owner, legal, lane);
}

// ------------------------------------------------------------------ search-space env (player's view)
struct Env {  // ReversiEnv in the searching player's coordinates: "black" = that player.  Uniform.
    raz_bb black, white;
    uint32_t np;      // 1 or 2
    uint32_t status;  // 0 running, else winner
    raz_bb legal;     // legal moves of the side to move (0 when done)
};

__device__ __forceinline__ void env_step(Env& e, int action) {
    raz_step_result r = bb_env_step(e.black, e.white, (int)e.np, action);
    e.black = r.black;
    e.white = r.white;
    e.np = r.player;
    e.status = r.status & RAZ_STATUS_WINNER_MASK;
    e.legal = r.legal;
}

// ------------------------------------------------------------------ select (agent/player.py:395-428)
// RANK space: lane r < k holds the statistics of the node's r-th legal move in ascending square order (the node's
// arrays as they lie in memory), the other lanes hold zeros.  The reference masks by the legal moves (`* legal`), so
// squares that are not legal never win the argmax; ascending rank is ascending square, so the first-maximum rule, the
// order of the Dirichlet samples (sample j belongs to the j-th legal square) and every sum below are the reference's.
// The node's P already holds normalize(P * legal) in float32 (player.py:404-413 gives the same
// vector at every visit of a node, so it is computed once, when the net's policy is stored).
// Returns the RANK of the chosen move.
__device__ int select_action(const raz_engine_dev& E, Regs& R, uint32_t g, double Wi, uint32_t Ni, float p32,
                             int k, uint32_t np, bool is_root, uint32_t game_id, int lane) {
    const raz_engine_config& c = E.cfg;
    const uint32_t bit = lane < k ? 1u : 0u;
    const uint32_t sumN = wave_sum_u32(Ni);
    double xx = sqrt((double)sumN);  // np.sqrt(np.sum(N)); correctly rounded on gfx950 (probe)
    if (xx < 1.0) xx = 1.0;           // max(xx_, 1)
    const double Nd = (double)Ni;
    double u;
    if (is_root && c.noise_eps > 0.0) {  // (1-eps) p + eps Dir(alpha), fresh at every root visit (:415-417)
        const unsigned long long tp = prof_now();
        const uint32_t ev = G32(R, GW(ev_dirichlet));
        double gam = 0.0;   // Gamma(alpha) sample j belongs to the j-th legal move = lane j
        if (c.dirichlet_alpha == 0.5) {
            // Box-Muller pairs: lane m < ceil(k/2) turns one Philox block into the two Gamma(1/2)
            // variates of legal moves 2m and 2m+1 (no rejection, no divergence).
            double g0 = 0.0, g1 = 0.0;
            if (2 * lane < k) raz_gamma_half_pair(c.seed, game_id, ev, (uint32_t)lane, g0, g1);
            const double s0 = __shfl(g0, lane >> 1), s1 = __shfl(g1, lane >> 1);
            if (bit) gam = (lane & 1) ? s1 : s0;
        } else {
            // Attempts of the rejection sampler are independent Philox blocks, so lane l evaluates attempt t = l / k of
            // sample j = l % k (up to 8 attempts per sample per round) and each legal move takes its sample's first
            // accepted attempt: one evaluation deep instead of the slowest lane's rejection count.
            int A = 64 / k;
            if (A > 8) A = 8;
            int myt = 0;
#pragma unroll
            for (int i = 1; i < 8; ++i) myt += (lane >= i * k) ? 1 : 0;
            const int myj = lane - myt * k;
            bool need = bit != 0;
            for (uint32_t round = 0;; ++round) {
                double X = 0.0;
                bool ok = false;
                if (lane < A * k) ok = raz_gamma_attempt(c.dirichlet_alpha, c.seed, game_id, ev, (uint32_t)myj, round * (uint32_t)A + (uint32_t)myt, X);
                const unsigned long long am = __ballot(ok);
                int src = -1;
                if (need) {
                    for (int t = 0; t < A; ++t) {
                        const int l = lane + k * t;
                        if ((am >> l) & 1ULL) {
                            src = l;
                            break;
                        }
                    }
                }
                const double got = __shfl(X, src < 0 ? lane : src);
                if (src >= 0) {
                    gam = got;
                    need = false;
                }
                if (__ballot(need) == 0ULL) break;
            }
        }
        double acc = 0.0;
        for (int j = 0; j < k; ++j) acc += lane_f64(gam, j);
        const double noise = bit ? gam / acc : 0.0;
        const float keep = (float)(1.0 - c.noise_eps);
        const double p64 = (double)(keep * p32) + c.noise_eps * noise;
        u = (c.c_puct * p64) * xx / (1.0 + Nd);
        S32(R, GW(ev_dirichlet), ev + 1);
        prof_add(E, g, 3, tp, lane);
    } else {
        const float cp = (float)c.c_puct;
        u = ((double)(cp * p32)) * xx / (1.0 + Nd);
    }
    const double q = Wi / (Nd + 1e-5);
    double v = (np == 1) ? (q + u + 1000.0) : (-q + u + 1000.0);
    v = v * (double)bit;
    return wave_argmax_nonneg_f64(v);  // == wave_argmax_f64(v, lane): v >= 0, first maximum
}

// P as select_action_q_and_u will use it: p = P * legal; if np.sum(p) > 0: p = p / np.sum(p)  (float32)
__device__ __forceinline__ float masked_normalised_prior(float pol, raz_bb legal, int lane, float* lds64) {
    float p32 = pol * (float)((legal >> lane) & 1ULL);
    const float sp = wave_np_sum_f32(p32, lane, lds64);
    if (sp > 0.0f) p32 = p32 / sp;
    return p32;
}

// ------------------------------------------------------------------ end-game solver
// lib/alt/reversi_solver_cython.pyx:40-127.  The reference's explicit-stack DFS returns a function of
// the position and the mode alone (see oracle/orc_solver.c): for the legal moves in ascending order
// [non-exact: stop once the best score is > 0], value = -f(child) / +f(child after a pass) / final
// disc difference, strict improvement keeps the first maximum.  Here: the same DFS, wave-uniform
// (scalar unit), frames in LDS (depth <= empties <= 14), with a per-game memo in HBM (positions
// with >= 4 empties; the memo only saves time, exactly as the reference's dict does).
struct SolverLDS {
    unsigned long long own[16], enemy[16], left[16];
    int best_move[16], best_score[16], paction[16], flip[16], fresh[16];
};

__device__ bool memo_find(const raz_engine_dev& E, uint32_t g, raz_bb own, raz_bb enemy, uint32_t exact, int lane,
                          int& move, int& score) {
    const raz_slot* tab = E.memo + (size_t)g * E.M;
    const uint32_t mask = E.M - 1;
    const uint32_t h = key_hash(own, enemy, 8u + exact);
    for (uint32_t r = 0; r < 64; r += RAZ_PROBE) {
        const raz_slot* s = tab + ((h + r + (uint32_t)(lane & (RAZ_PROBE - 1))) & mask);
        const raz_bb sb = s->black, sw = s->white;
        const uint32_t it = s->idx_tag;
        const bool used = (it >> 31) != 0;
        const bool match = used && sb == own && sw == enemy && ((it >> 30) & 1u) == exact;
        const unsigned long long mm = __ballot(match) & 0xffffULL;
        const unsigned long long em = __ballot(!used) & 0xffffULL;
        if (mm) {
            const uint32_t v = lane_u32(it, __ffsll((long long)mm) - 1);
            move = (int)((v >> 8) & 0xffu) - 1;
            score = (int)(v & 0xffu) - 128;
            return true;
        }
        if (em) return false;
    }
    return false;
}

__device__ void memo_put(const raz_engine_dev& E, uint32_t g, raz_bb own, raz_bb enemy, uint32_t exact, int move,
                         int score, int lane) {
    raz_slot* tab = E.memo + (size_t)g * E.M;
    const uint32_t mask = E.M - 1;
    const uint32_t h = key_hash(own, enemy, 8u + exact);
    for (uint32_t r = 0; r < 64; r += RAZ_PROBE) {
        const uint32_t si = (h + r + (uint32_t)(lane & (RAZ_PROBE - 1))) & mask;
        const bool used = (tab[si].idx_tag >> 31) != 0;
        const unsigned long long em = __ballot(!used) & 0xffffULL;
        if (em) {
            if (lane == 0) {
                raz_slot* s = tab + ((h + r + (uint32_t)(__ffsll((long long)em) - 1)) & mask);
                s->black = own;
                s->white = enemy;
                s->idx_tag = 0x80000000u | (exact << 30) | ((uint32_t)(move + 1) << 8) | (uint32_t)(score + 128);
            }
            wave_sync();
            return;
        }
    }  // 64 occupied slots in a row: the memo is (locally) full; skipping the insert only costs time
}

// ReversiSolver.solve for the side to move (own, enemy).  Returns false for the reference's
// (None, None) (no legal move at the root: never the case for a running game).
__device__ bool solver_solve(const raz_engine_dev& E, uint32_t g, int lane, raz_bb own0, raz_bb enemy0, uint32_t exact,
                             SolverLDS* S, int& out_move, int& out_score) {
    int depth = 0;
    S->own[0] = own0;
    S->enemy[0] = enemy0;
    S->left[0] = bb_legal_moves(own0, enemy0);
    S->best_move[0] = -1;
    S->best_score[0] = -100;
    S->paction[0] = -1;
    S->flip[0] = 0;
    S->fresh[0] = 1;
    for (;;) {
        wave_sync();
        const raz_bb own = uni(S->own[depth]), enemy = uni(S->enemy[depth]);
        raz_bb left = uni(S->left[depth]);
        int best_move = uni(S->best_move[depth]), best_score = uni(S->best_score[depth]);
        const bool big = bb_popcount(~(own | enemy)) >= 4;
        int rm = 0, rs = 0;
        bool done = false;
        if (uni(S->fresh[depth])) {
            S->fresh[depth] = 0;
            if (big && memo_find(E, g, own, enemy, exact, lane, rm, rs)) done = true;
        }
        if (!done && (left == 0 || (!exact && best_score > 0))) {
            if (big) memo_put(E, g, own, enemy, exact, best_move, best_score, lane);
            rm = best_move;
            rs = best_score;
            done = true;
        }
        if (done) {
            if (depth == 0) {
                out_move = rm;
                out_score = rs;
                return rm >= 0;
            }
            const int pa = uni(S->paction[depth]);
            const int v = uni(S->flip[depth]) ? -rs : rs;
            --depth;
            if (uni(S->best_score[depth]) < v) {
                S->best_move[depth] = pa;
                S->best_score[depth] = v;
            }
            continue;
        }
        const int a = __ffsll((long long)left) - 1;
        left &= left - 1;
        S->left[depth] = left;
        const raz_bb flipped = bb_calc_flip(a, own, enemy);
        const raz_bb nown = (own ^ flipped) | (1ULL << a), nenemy = enemy ^ flipped;
        const raz_bb l1 = bb_legal_moves(nenemy, nown);
        if (l1 || bb_legal_moves(nown, nenemy)) {
            const bool turn_passes = l1 == 0;  // the opponent has no move: same side again
            ++depth;
            S->own[depth] = turn_passes ? nown : nenemy;
            S->enemy[depth] = turn_passes ? nenemy : nown;
            S->left[depth] = turn_passes ? bb_legal_moves(nown, nenemy) : l1;
            S->best_move[depth] = -1;
            S->best_score[depth] = -100;
            S->paction[depth] = a;
            S->flip[depth] = turn_passes ? 0 : 1;
            S->fresh[depth] = 1;
        } else {
            const int score = bb_popcount(nown) - bb_popcount(nenemy);
            if (best_score < score) {
                S->best_move[depth] = a;
                S->best_score[depth] = score;
            }
        }
    }
}

// ------------------------------------------------------------------ backup of the previous leaf
// Leaf node + its colour-mirrored node for a leaf that is being expanded (prior = the net's policy)
// or was solved (prior = one-hot): create / update them and cross-link.  Returns false when out of
// space.  `used` / `count` are the running pool counters (8-byte units / nodes); `prior` is in SQUARE space.
__device__ bool place_leaf(const raz_engine_dev& E, Regs& R, uint32_t g, int lane, uint32_t owner, uint32_t np,
                           raz_bb kb, raz_bb kw, raz_bb lg, uint32_t new_tag_bits, float prior, bool with_mirror,
                           int depth, uint32_t& node, uint32_t& mirror, uint32_t& used, uint32_t& count) {
    node = G32(R, GW(leaf_node));
    mirror = RAZ_NO_NODE;
    const uint32_t tagkey = np | (owner << 2);
    const int L = __popcll(lg);
    const bool on = (lg >> lane) & 1ULL;
    const int rk = rank_of(lg, lane);
    const uint32_t old_tag = G32(R, GW(leaf_tag)), old_mirror = G32(R, GW(leaf_mirror));   // (cross-lane reads stay outside lane-0 branches)
    if (node == RAZ_NO_NODE) {  // first arrival at this position: create it in the slot select found
        uint32_t slot = G32(R, GW(leaf_slot));
        if (slot == 0xfffffffeu) slot = table_find(E, g, kb, kw, tagkey, lane).slot;  // table rebuilt by k_gc
        node = slot == 0xffffffffu ? RAZ_NO_NODE : pool_take(E, L, used, count);
        if (node == RAZ_NO_NODE) {
            flag_error(R, (slot == 0xffffffffu) ? RAZ_ERR_TABLE_FULL : RAZ_ERR_POOL_FULL);
            return false;
        }
        node_init(E, g, node, count - 1, slot, kb, kw, tagkey | new_tag_bits, lg, RAZ_NO_NODE, prior, lane);
        if (depth > 0) {  // link the parent's edge to it
            const uint32_t parent = lane_u32(R.pnode, depth - 1);
            const uint32_t pa = lane_u32(R.pact, depth - 1);
            if (lane == 0) node_child(node_ptr(E, g, parent), link_L(parent))[pa >> 8] = node;
        }
    } else {  // existing node: store the prior (and the expanded flag)
        unsigned char* p = node_ptr(E, g, node);
        if (on) node_P(p, L)[rk] = prior;
        if (new_tag_bits && lane == 0) node_hdr(p)->tag = old_tag | new_tag_bits;
        mirror = old_mirror;
    }
    if (!with_mirror) {
        mirror = RAZ_NO_NODE;
        return true;
    }
    if (mirror == RAZ_NO_NODE) {  // var_p[another_side_key] = leaf_p (:324): the mirror key may be new too
        wave_sync();
        const Found f = table_find(E, g, kw, kb, (3 - np) | (owner << 2), lane);
        if (f.found) {
            mirror = f.node;
            if (on) node_P(node_ptr(E, g, mirror), L)[rk] = prior;
        } else {
            mirror = f.slot == 0xffffffffu ? RAZ_NO_NODE : pool_take(E, L, used, count);
            if (mirror == RAZ_NO_NODE)
                flag_error(R, (f.slot == 0xffffffffu) ? RAZ_ERR_TABLE_FULL : RAZ_ERR_POOL_FULL);
            else   // (the mirror key has the same side's moves on the same squares: same legal mask, same L)
                node_init(E, g, mirror, count - 1, f.slot, kw, kb, (3 - np) | (owner << 2), lg, node, prior, lane);
        }
        if (mirror != RAZ_NO_NODE && lane == 0) {
            node_hdr(node_ptr(E, g, node))->mirror = mirror;
            node_hdr(node_ptr(E, g, mirror))->mirror = node;
        }
    } else {
        if (on) node_P(node_ptr(E, g, mirror), L)[rk] = prior;
    }
    return true;
}

// Everything the backup needs is in registers (control block, path, the net's answer), so its
// memory work is: the (N, W) cells of the path levels (lane d = level d, issued first), the table
// probe for the mirror key of a brand-new position, and stores.
// PAR (parallel_search_num > 1): the virtual loss was STORED when the edge was taken (other simulations
// of the game read it meanwhile), so the return path only adds "-vl + 1" / "vlw + leaf_v" (:276-277).
template <bool PAR>
__device__ void backup_leaf(const raz_engine_dev& E, Regs& R, uint32_t g, uint32_t pl, int lane, float* lds64) {
    const raz_engine_config& c = E.cfg;
    const uint32_t kind = G32(R, GW(leaf_kind));
    if (kind == RAZ_LEAF_NONE) return;
    const uint32_t owner = c.share_mtcs_info ? 0u : pl;
    const int depth = (int)G32(R, GW(depth));
    // the path cells: every level is an independent (node, rank of the action) cell; a node and its colour-mirrored
    // node have the same legal mask, hence the same array length and the same rank for an action
    const uint32_t my_node = R.pnode, my_mirror = R.pmirror, my_pa = R.pact;
    const uint32_t a = my_pa >> 8;
    const uint32_t m = c.mirror_updates ? my_mirror : RAZ_NO_NODE;
    const int Lp = link_L(my_node);
    uint32_t n0 = 0, n1 = 0;
    double w0 = 0.0, w1 = 0.0;
    unsigned char *p = nullptr, *q = nullptr;
    if (lane < depth) {  // issued first: in flight while the leaf is placed
        p = node_ptr(E, g, my_node);
        q = node_ptr(E, g, m == RAZ_NO_NODE ? my_node : m);
        n0 = node_N(p, Lp)[a];
        w0 = node_W(p, Lp)[a];
        n1 = node_N(q, Lp)[a];
        w1 = node_W(q, Lp)[a];
    }
    double leaf_v;
    const unsigned long long te = prof_now();
    if (kind == RAZ_LEAF_EXPAND) {  // expand_and_evaluate (:283-327), second half
        const uint32_t np = G32(R, GW(leaf_np)), sym = G32(R, GW(leaf_sym));
        const raz_bb lg = G64(R, GW(leaf_legal)), kb = G64(R, GW(leaf_b)), kw = G64(R, GW(leaf_w));
        leaf_v = (double)R.val;          // float(leaf_v)
        if (np == 2) leaf_v = -leaf_v;   // :259-262
        // the net saw T(board); its policy q is over T-squares, so p[s] = q[T(s)]
        const float pol = __shfl(R.pol_raw, bb_d4_square(lane, (sym >> 2) & 1, sym & 3));
        const float pn = masked_normalised_prior(pol, lg, lane, lds64);
        uint32_t used = G32(R, GW(pool_used)), count = G32(R, GW(node_count));
        uint32_t node, mirror;
        place_leaf(E, R, g, lane, owner, np, kb, kw, lg, 16u << pl, pn, c.mirror_updates != 0, depth, node, mirror, used, count);
        S32(R, GW(pool_used), used);
        S32(R, GW(node_count), count);
    } else if (kind == RAZ_LEAF_SOLVED) {  // in-simulation solver hit (:239-251): the key and its mirror get
        // N += 1, W +-= sign(score), P = one-hot; the key is NOT marked expanded
        const uint32_t np = G32(R, GW(leaf_np)), act = G32(R, GW(leaf_action));
        const raz_bb lg = G64(R, GW(leaf_legal)), kb = G64(R, GW(leaf_b)), kw = G64(R, GW(leaf_w));
        leaf_v = (double)raz_bits_to_f32(G32(R, GW(leaf_term_v)));  // sign(score) in the searching player's view
        const float onehot = lane == (int)act ? 1.0f : 0.0f;
        uint32_t used = G32(R, GW(pool_used)), count = G32(R, GW(node_count));
        uint32_t node, mirror;
        // (:248-250) writes to the mirror key are dead without a shared tree
        const bool ok = place_leaf(E, R, g, lane, owner, np, kb, kw, lg, 0u, onehot, c.mirror_updates != 0, depth, node, mirror, used, count);
        if (ok) {
            wave_sync();
            if (lane == 0) {
                const int L = __popcll(lg), ra = rank_of(lg, (int)act);
                unsigned char* pp = node_ptr(E, g, node);
                node_N(pp, L)[ra] += 1u;
                node_W(pp, L)[ra] = node_W(pp, L)[ra] + leaf_v;
                if (mirror != RAZ_NO_NODE) {
                    unsigned char* qq = node_ptr(E, g, mirror);
                    node_N(qq, L)[ra] += 1u;
                    node_W(qq, L)[ra] = node_W(qq, L)[ra] - leaf_v;
                }
            }
        }
        S32(R, GW(pool_used), used);
        S32(R, GW(node_count), count);
    } else {
        leaf_v = (double)raz_bits_to_f32(G32(R, GW(leaf_term_v)));
    }
    prof_add(E, g, 7, te, lane);
    if (lane < depth) {  // N += vl; W -= vlw; ...; N += -vl + 1; W += vlw + leaf_v  (:270-277)
        const double vl = (double)c.virtual_loss;
        const uint32_t npd = (my_pa >> 6) & 3u;
        const double vlw = npd == 1 ? vl : -vl;
        if (PAR) {
            node_N(p, Lp)[a] = n0 + 1u - (uint32_t)c.virtual_loss;
            node_W(p, Lp)[a] = w0 + (vlw + leaf_v);
        } else {
            node_N(p, Lp)[a] = n0 + 1u;
            node_W(p, Lp)[a] = (w0 - vlw) + (vlw + leaf_v);
        }
        if (m != RAZ_NO_NODE) {  // another_side_counter_key (:279-280); exists since the node's expansion
            node_N(q, Lp)[a] = n1 + 1u;
            node_W(q, Lp)[a] = w1 - leaf_v;
        }
    }
    S32(R, GW(leaf_kind), RAZ_LEAF_NONE);
    S32(R, GW(sims_left), G32(R, GW(sims_left)) - 1u);
    S32(R, GW(move_sims), G32(R, GW(move_sims)) + 1u);
    ADD64(R, GW(sims), 1ULL);
    wave_sync();
}

// Record the chosen move and play it on the real board (worker/self_play.py:155-162).
__device__ void finalize_move(const raz_engine_dev& E, Regs& R, uint32_t g, int lane, uint32_t player, raz_bb rb, raz_bb rw,
                              raz_bb own, raz_bb enemy, int turn, int final_action, bool has_row, bool solved,
                              double n_action, double q_action, uint32_t loops, uint32_t Ni, double Wi) {
    // record the ply (rows + GGF are produced on the host from this)
    const uint32_t ply = G32(R, GW(n_plies));
    if (ply >= E.max_plies) {
        flag_error(R, RAZ_ERR_RECORDS_FULL);
        S32(R, GW(phase), RAZ_PHASE_DONE);
        return;
    }
    const size_t ri = (size_t)g * E.max_plies + ply;
    E.rec_n[ri * 64 + lane] = Ni;
    if (E.rec_w) E.rec_w[ri * 64 + lane] = Wi;
    const uint32_t move_sims = G32(R, GW(move_sims));   // (cross-lane reads stay outside lane-0 branches)
    if (lane == 0) {
        raz_ply_header h;
        h.own = own;
        h.enemy = enemy;
        h.n = final_action >= 0 ? n_action : 0.0;
        h.q = final_action >= 0 ? q_action : 0.0;
        h.action = (int8_t)final_action;
        h.player = (uint8_t)player;
        h.turn = (uint8_t)turn;
        h.has_row = has_row ? 1 : 0;
        h.sims = move_sims;
        h.loops = loops;
        h.flags = solved ? 1u : 0u;
        E.rec[ri] = h;
    }
    S32(R, GW(n_plies), ply + 1);
    // env.step(action) on the real board (worker/self_play.py:162)
    raz_step_result r = bb_env_step(rb, rw, (int)player, final_action < 0 ? RAZ_ACTION_RESIGN : final_action);
    S64(R, GW(root_black), r.black);
    S64(R, GW(root_white), r.white);
    S32(R, GW(player), r.player);
    S32(R, GW(status), r.status);
    S32(R, GW(loops_done), 0);
    S32(R, GW(move_sims), 0);
    // one-move mode (ReversiPlayer facade): the slot idles after its move instead of playing on
    S32(R, GW(phase), G32(R, GW(one_move)) ? RAZ_PHASE_IDLE : (r.status ? RAZ_PHASE_DONE : RAZ_PHASE_NEW_MOVE));
}

// ------------------------------------------------------------------ per-move controller
// action_with_evaluation (:82-134) after a search (or the turn-0 bypass) has finished, then
// SelfPlayWorker.start_game's env.step (worker/self_play.py:155-162).  Returns with the game either
// searching again (phase SEARCH, sims_left > 0), waiting for a new move (phase NEW_MOVE) or DONE.
__device__ void decide_move(const raz_engine_dev& E, Regs& R, uint32_t g, int lane) {
    const raz_engine_config& c = E.cfg;
    const uint32_t player = G32(R, GW(player));
    const uint32_t pl = player - 1;
    const raz_bb rb = G64(R, GW(root_black)), rw = G64(R, GW(root_white));
    const raz_bb own = player == 1 ? rb : rw, enemy = player == 1 ? rw : rb;
    const int turn = bb_popcount(own) + bb_popcount(enemy) - 4;
    const uint32_t game_id = G32(R, GW(game_id));
    const uint32_t node = G32(R, GW(root_node));
    if (node == RAZ_NO_NODE) {
        S32(R, GW(phase), RAZ_PHASE_DONE);
        return;
    }
    unsigned char* p = node_ptr(E, g, node);
    uint32_t Ni;   // the root's statistics by SQUARE (var_n[key][action], var_w[key][action]: zero off the legal moves)
    double Wi;
    node_read_squares(p, link_L(node), bb_legal_moves(own, enemy), lane, Wi, Ni);
    const double Nd = (double)Ni;
    const double q = Wi / (Nd + 1e-5);
    const uint32_t sumN = wave_sum_u32(Ni);
    // calc_policy (:366-385)
    double policy;
    const int amax_n = wave_argmax_f64(Nd, lane);
    if (turn < c.change_tau_turn)
        policy = Nd / (double)sumN;
    else
        policy = (lane == amax_n) ? 1.0 : 0.0;
    // np.random.choice(range(64), p=policy) (:112): cdf = cumsum; cdf /= cdf[-1]; searchsorted right
    double acc = 0.0, cdf = 0.0;
    for (int i = 0; i < 64; ++i) {
        acc += lane_f64(policy, i);
        if (i == lane) cdf = acc;
    }
    cdf = cdf / acc;
    const uint32_t ev = G32(R, GW(ev_choice));
    double d0, d1;
    raz_rng_pair(c.seed, game_id, RAZ_RNG_CHOICE, ev, 0, 0, d0, d1);
    int action = __popcll(__ballot(cdf <= d0));
    if (action > 63) action = 63;
    S32(R, GW(ev_choice), ev + 1);
    // re-thinking rule (:113-118)
    const int abv = wave_argmax_f64(q + (Ni > 0 ? 100.0 : 0.0), lane);
    const double q_action = lane_f64(q, action), q_abv = lane_f64(q, abv);
    const double n_action = lane_f64(Nd, action);
    const double value_diff = q_action - q_abv;
    const uint32_t loops = G32(R, GW(loops_done)) + 1;
    const bool stop = (turn <= c.start_rethinking_turn) ||
                      (value_diff > -0.01 && n_action >= (double)c.required_visit_to_decide_action) ||
                      ((int)loops >= c.thinking_loop);
    if (!stop) {  // another thinking loop on the same root (tree and N are kept)
        S32(R, GW(loops_done), loops);
        S32(R, GW(sims_left), G32(R, GW(sims_per_move)));
        S32(R, GW(phase), RAZ_PHASE_SEARCH);
        return;
    }
    // resignation (:123-130)
    int final_action = action;
    bool has_row = true;
    const uint32_t rmode = G32(R, GW(resign_mode));   // per-game rule of a game started by raz_engine_harvest, else the engine's
    const bool has_thr = rmode ? rmode == 1u : c.has_resign_threshold != 0;
    if (has_thr) {
        const double thr = rmode == 1u ? __longlong_as_double((long long)G64(R, GW(resign_thr))) : c.resign_threshold;
        const double mx = wave_max_f64(q - (Ni == 0 ? 10.0 : 0.0));
        if (mx <= thr) {
            R.cw = writelane_r(R.cw, 1u, GW(resigned) + (int)pl, lane);
            if (G32(R, GW(enable_resign)) && turn >= c.allowed_resign_turn) {
                final_action = -1;
                has_row = false;
            }
        }
    }
    finalize_move(E, R, g, lane, player, rb, rw, own, enemy, turn, final_action, has_row, false, n_action, q_action, loops, Ni, Wi);
}

// Start the mover's move: find/create its root node; turn 0 -> bypass_first_move (:143-148),
// else arm a search.
template <bool SOLVER>
__device__ void begin_move(const raz_engine_dev& E, Regs& R, uint32_t g, int lane, SolverLDS* S) {
    const raz_engine_config& c = E.cfg;
    const uint32_t player = G32(R, GW(player));
    const uint32_t pl = player - 1;
    const uint32_t owner = c.share_mtcs_info ? 0u : pl;
    const raz_bb rb = G64(R, GW(root_black)), rw = G64(R, GW(root_white));
    const raz_bb own = player == 1 ? rb : rw, enemy = player == 1 ? rw : rb;
    const int turn = bb_popcount(own) + bb_popcount(enemy) - 4;
    const raz_bb legal = bb_legal_moves(own, enemy);
    const uint32_t node = node_get(E, R, g, own, enemy, 1, owner, legal, lane);
    S32(R, GW(root_node), node);
    const int L = __popcll(legal), rk = rank_of(legal, lane);
    const bool on = (legal >> lane) & 1ULL;
    if (SOLVER && c.use_solver_turn && turn >= c.use_solver_turn && node != RAZ_NO_NODE) {  // action_by_searching (:100-103,150-161)
        int sm, ss;
        if (solver_solve(E, g, lane, own, enemy, 1u, S, sm, ss)) {
            unsigned char* p = node_ptr(E, g, node);
            const double sg = ss > 0 ? 1.0 : (ss < 0 ? -1.0 : 0.0);
            if (on) node_P(p, L)[rk] = lane == sm ? 1.0f : 0.0f;
            uint32_t Ni;
            double Wi;
            node_read_squares(p, L, legal, lane, Wi, Ni);
            if (lane == sm) {
                Ni = 999u;
                Wi = sg * 999.0;
                node_N(p, L)[rk] = Ni;
                node_W(p, L)[rk] = Wi;
            }
            wave_sync();
            finalize_move(E, R, g, lane, player, rb, rw, own, enemy, turn, sm, false, true, 999.0, sg, 0u, Ni, Wi);
            return;
        }
    }
    if (turn > 0) {
        S32(R, GW(sims_left), G32(R, GW(sims_per_move)));
        S32(R, GW(phase), RAZ_PHASE_SEARCH);
    } else {
        if (node != RAZ_NO_NODE) {
            unsigned char* p = node_ptr(E, g, node);
            const int cnt = bb_popcount(legal);
            if (on) node_P(p, L)[rk] = (float)((double)((legal >> lane) & 1ULL) / (double)cnt);
            if (on && rk == 0) {   // the first legal move
                node_N(p, L)[0] = 1u;
                node_W(p, L)[0] = 0.0;
            }
        }
        S32(R, GW(sims_left), 0u);
        S32(R, GW(phase), RAZ_PHASE_SEARCH);  // "search" of zero simulations: decide immediately
    }
    wave_sync();
}

// ------------------------------------------------------------------ descent to the next leaf
// PAR (k_tree_par): the descent may start at the node a sleeping simulation stood on (start_node /
// start_depth, `polling`: the solver look of :237-251 lies behind it), every edge taken gets its virtual
// loss stored at once (:270-271), a leaf being expanded is flagged in its node's tag (now_expanding,
// :294) - a brand-new position gets its node right here for that - and a descent that meets such a
// flag stops there (RAZ_LEAF_PARKED, :253-254).  nn_index: the slot of the leaf exchange arrays.
template <bool SOLVER, bool PAR>
__device__ void select_leaf(const raz_engine_dev& E, Regs& R, uint32_t g, int lane, SolverLDS* S, uint32_t nn_index,
                            uint32_t start_node, int start_depth, bool polling) {
    const raz_engine_config& c = E.cfg;
    const uint32_t player = G32(R, GW(player));
    const uint32_t pl = player - 1;
    const uint32_t owner = c.share_mtcs_info ? 0u : pl;
    const uint32_t game_id = G32(R, GW(game_id));
    const raz_bb rb = G64(R, GW(root_black)), rw = G64(R, GW(root_white));
    Env env;  // ReversiEnv().update(own, enemy, Player.black) (:209)
    env.black = player == 1 ? rb : rw;
    env.white = player == 1 ? rw : rb;
    env.np = 1;
    env.status = 0;
    env.legal = 0;
    int depth = PAR ? start_depth : 0;
    uint32_t kind = RAZ_LEAF_NONE;
    uint32_t node = PAR ? start_node : G32(R, GW(root_node));  // always exists (begin_move)
    uint32_t leaf_node = RAZ_NO_NODE, leaf_slot = 0xffffffffu, leaf_tag = 0, leaf_mirror = RAZ_NO_NODE;
    raz_bb leaf_legal = 0;
    int solved_action = 0;
    float term_v = 0.0f;
    const int t_insim = SOLVER ? c.use_solver_turn_in_simulation : 0;
    if (node == RAZ_NO_NODE) {
        S32(R, GW(phase), RAZ_PHASE_DONE);
        return;
    }
    for (;;) {
        // one round trip: header (broadcast) + the node's four arrays, lane r < L holding the r-th legal move (the link
        // carries L, so the arrays' addresses do not wait for the header).  The header carries the position itself, so
        // following a linked edge needs no move generation: flips and legal moves are computed once, when an edge is
        // first taken.
        unsigned char* p = node_ptr(E, g, node);
        const int L = link_L(node);
        const raz_node_hdr* hp = node_hdr(p);
        const raz_bb hb = hp->black, hw = hp->white, legal = hp->legal;
        const uint32_t tag = hp->tag, hmirror = hp->mirror;
        const bool have = lane < L;
        const double Wi = have ? node_W(p, L)[lane] : 0.0;
        const uint32_t Ni = have ? node_N(p, L)[lane] : 0u;
        const float Pi = have ? node_P(p, L)[lane] : 0.0f;
        const uint32_t Ci = have ? node_child(p, L)[lane] : 0u;
        if (RAZ_PROF_ON(E)) {
            const unsigned long long tl = prof_now();
#ifndef RAZ_WAVE_EMU
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            prof_add(E, g, 6, tl, lane);
        }
        env.black = uni(hb);
        env.white = uni(hw);
        env.np = uni(tag) & 3u;
        env.legal = uni(legal);
        if (SOLVER && t_insim && !(PAR && polling) && bb_popcount(env.black | env.white) - 4 >= t_insim) {  // solver inside simulations (:237-251)
            const raz_bb so = env.np == 1 ? env.black : env.white, se = env.np == 1 ? env.white : env.black;
            int sm, ss;
            if (solver_solve(E, g, lane, so, se, 0u, S, sm, ss) && sm != 0) {  // `if action:` ignores square 0
                if (env.np != 1) ss = -ss;
                kind = RAZ_LEAF_SOLVED;
                solved_action = sm;
                term_v = ss > 0 ? 1.0f : (ss < 0 ? -1.0f : 0.0f);
                leaf_node = node;
                leaf_legal = env.legal;
                leaf_tag = uni(tag);
                leaf_mirror = uni(hmirror);
                break;
            }
        }
        if (PAR) {
            polling = false;
            if ((uni(tag) >> (6 + pl)) & 1u) {  // while key in self.now_expanding: await asyncio.sleep(...) (:253-254)
                kind = RAZ_LEAF_PARKED;
                leaf_node = node;
                break;
            }
        }
        if (!((uni(tag) >> (4 + pl)) & 1u)) {  // key not in this player's `expanded` (:257)
            kind = RAZ_LEAF_EXPAND;
            leaf_node = node;
            leaf_legal = env.legal;
            leaf_tag = uni(tag);
            leaf_mirror = uni(hmirror);
            if (PAR && lane == 0) node_hdr(p)->tag = leaf_tag | (64u << pl);  // now_expanding.add(key) (:294)
            break;
        }
        if (depth >= 64) {
            flag_error(R, RAZ_ERR_PATH_FULL);
            break;
        }
        const int r = select_action(E, R, g, Wi, Ni, Pi, L, env.np, depth == 0, game_id, lane);   // rank of the move
        const int a = square_of_rank(env.legal, r, lane);
        if (PAR && lane == r) {  // var_n[key][action_t] += virtual_loss; var_w[key][action_t] -= virtual_loss_for_w (:270-271)
            const double vl = (double)c.virtual_loss;
            node_N(p, L)[r] = Ni + (uint32_t)c.virtual_loss;
            node_W(p, L)[r] = Wi - (env.np == 1 ? vl : -vl);
        }
        R.pnode = writelane_r(R.pnode, node, depth, lane);
        R.pmirror = writelane_r(R.pmirror, hmirror, depth, lane);
        R.pact = writelane_r(R.pact, (uint32_t)a | (env.np << 6) | ((uint32_t)r << 8), depth, lane);
        ++depth;
        const uint32_t child = lane_u32(Ci, r);
        if (child & 0x80000000u) {  // edge known to end the game: env.done (:226-232)
            const uint32_t w = child & 3u;
            kind = RAZ_LEAF_TERMINAL;
            term_v = w == RAZ_WIN_BLACK ? 1.0f : (w == RAZ_WIN_WHITE ? -1.0f : 0.0f);
            break;
        }
        if (child) {
            node = child;
            continue;
        }
        // first time along this edge: play the move
        env_step(env, a);
        if (env.status) {  // env.done (:226-232); remember the result on the edge
            kind = RAZ_LEAF_TERMINAL;
            term_v = env.status == RAZ_WIN_BLACK ? 1.0f : (env.status == RAZ_WIN_WHITE ? -1.0f : 0.0f);
            if (lane == 0) node_child(p, L)[r] = 0x80000000u | env.status;
            break;
        }
        // the position may already exist (transposition / mirror write)
        const unsigned long long tq = prof_now();
        const Found f = table_find(E, g, env.black, env.white, env.np | (owner << 2), lane);
        prof_add(E, g, 4, tq, lane);
        if (f.found) {
            if (lane == 0) node_child(p, L)[r] = f.node;
            node = f.node;
            continue;
        }
        leaf_slot = f.slot;  // brand-new position: created at backup time in the slot found here
        leaf_legal = env.legal;
        kind = RAZ_LEAF_EXPAND;
        if (SOLVER && t_insim && bb_popcount(env.black | env.white) - 4 >= t_insim) {  // (:237-251) on a first arrival
            const raz_bb so = env.np == 1 ? env.black : env.white, se = env.np == 1 ? env.white : env.black;
            int sm, ss;
            if (solver_solve(E, g, lane, so, se, 0u, S, sm, ss) && sm != 0) {
                if (env.np != 1) ss = -ss;
                kind = RAZ_LEAF_SOLVED;
                solved_action = sm;
                term_v = ss > 0 ? 1.0f : (ss < 0 ? -1.0f : 0.0f);
            }
        }
        if (PAR && kind == RAZ_LEAF_EXPAND) {  // the key enters now_expanding: it needs a node to carry the flag
            uint32_t used = G32(R, GW(pool_used)), count = G32(R, GW(node_count));
            const uint32_t fresh = f.slot == 0xffffffffu ? RAZ_NO_NODE : pool_take(E, __popcll(env.legal), used, count);
            if (fresh == RAZ_NO_NODE) {
                flag_error(R, (f.slot == 0xffffffffu) ? RAZ_ERR_TABLE_FULL : RAZ_ERR_POOL_FULL);
                kind = RAZ_LEAF_NONE;
                break;
            }
            const uint32_t tagkey = env.np | (owner << 2);
            node_init(E, g, fresh, count - 1, f.slot, env.black, env.white, tagkey | (64u << pl), env.legal, RAZ_NO_NODE, 0.0f, lane);
            if (lane == 0) node_child(p, L)[r] = fresh;
            S32(R, GW(pool_used), used);
            S32(R, GW(node_count), count);
            leaf_node = fresh;
            leaf_tag = tagkey;
        }
        break;
    }
    R.path_dirty = 1u;
    if (kind == RAZ_LEAF_EXPAND || kind == RAZ_LEAF_SOLVED) {
        S64(R, GW(leaf_b), env.black);
        S64(R, GW(leaf_w), env.white);
        S64(R, GW(leaf_legal), leaf_legal);
        S32(R, GW(leaf_node), leaf_node);
        S32(R, GW(leaf_slot), leaf_slot);
        S32(R, GW(leaf_tag), leaf_tag);
        S32(R, GW(leaf_mirror), leaf_mirror);
        S32(R, GW(leaf_np), env.np);
    }
    if (kind == RAZ_LEAF_EXPAND) {  // expand_and_evaluate (:283-311), first half
        const uint32_t ev = G32(R, GW(ev_expand));
        double d0, d1;
        raz_rng_pair(c.seed, game_id, RAZ_RNG_EXPAND, ev, 0, 0, d0, d1);
        const int flip = d0 < 0.5 ? 1 : 0;   // random() < 0.5
        const int rot = (int)(d1 * 4.0);     // int(random() * 4)
        const raz_bb tb = bb_d4_apply(env.black, flip, rot), tw = bb_d4_apply(env.white, flip, rot);
        S32(R, GW(ev_expand), ev + 1);
        S32(R, GW(leaf_sym), (uint32_t)(flip * 4 + rot));
        if (lane == 0) {
            E.nn_own[PAR ? nn_index : g] = env.np == 1 ? tb : tw;   // planes from the side to move's view (:309)
            E.nn_enemy[PAR ? nn_index : g] = env.np == 1 ? tw : tb;
        }
        ADD64(R, GW(leaves), 1ULL);
        R.nn = 1u;
    }
    if (kind == RAZ_LEAF_SOLVED) S32(R, GW(leaf_action), (uint32_t)solved_action);
    if (PAR && kind == RAZ_LEAF_PARKED) S32(R, GW(sim_parked), leaf_node);
    S32(R, GW(leaf_term_v), __float_as_uint(term_v));
    S32(R, GW(leaf_kind), kind);
    S32(R, GW(depth), (uint32_t)depth);
    ADD64(R, GW(selections), (raz_bb)(depth - (PAR ? start_depth : 0)));
    wave_sync();
}

// The path of the simulation in flight (lane d = level d).  Paths are short (2-6 levels; the mean is ~2.5), so only the
// first 16 levels travel with the control block; the rest - and only the levels in use are ever stored - is fetched in the
// rare case of a deeper path (one extra round trip).
#ifndef RAZ_PATH_EAGER
#define RAZ_PATH_EAGER 16   // (the wave-emulator build sets 2, so that its tiny games exercise the deep-path fetch all the time)
#endif
constexpr int kPathEager = RAZ_PATH_EAGER;
__device__ __forceinline__ void path_load(const raz_engine_dev& E, Regs& R, size_t row, int lane, bool eager_only) {
    const bool take = !eager_only || lane < kPathEager;
    R.pnode = take ? E.path_node[row * 64 + lane] : 0u;
    R.pmirror = take ? E.path_mirror[row * 64 + lane] : 0u;
    R.pact = take ? (uint32_t)E.path_act[row * 64 + lane] : 0u;
}
__device__ __forceinline__ void path_load_rest(const raz_engine_dev& E, Regs& R, size_t row, int lane) {
    const int depth = (int)G32(R, GW(depth));   // (of the block now in R.cw: the game's, or the simulation slot's just loaded)
    if (depth > kPathEager && lane >= kPathEager) {
        R.pnode = E.path_node[row * 64 + lane];
        R.pmirror = E.path_mirror[row * 64 + lane];
        R.pact = (uint32_t)E.path_act[row * 64 + lane];
    }
}
__device__ __forceinline__ void path_store(const raz_engine_dev& E, const Regs& R, size_t row, int lane) {
    const int depth = (int)G32(R, GW(depth));
    if (lane < depth) {
        E.path_node[row * 64 + lane] = R.pnode;
        E.path_mirror[row * 64 + lane] = R.pmirror;
        E.path_act[row * 64 + lane] = (uint16_t)R.pact;
    }
}

// ------------------------------------------------------------------ the tree kernel
// SOLVER = false compiles the end-game solver (and its LDS frames) out: the common case, and the
// bench configuration.
template <bool SOLVER>
__global__ __launch_bounds__(64) void k_tree(raz_engine_dev E, uint32_t g0, uint32_t count) {
    if (blockIdx.x >= count) return;
    __shared__ float lds64[64];
    __shared__ SolverLDS slds_store;
    SolverLDS* slds_p = SOLVER ? &slds_store : nullptr;
    const uint32_t g = g0 + blockIdx.x;
    const int lane = threadIdx.x;
    if (g >= E.B) return;
    // ONE round trip: control block, path, the net's answer for the leaf of the previous launch
    uint32_t* gw = (uint32_t*)(E.game + g);
    Regs R;
    R.cw = gw[lane];
    path_load(E, R, (size_t)g, lane, true);
    R.pol_raw = E.nn_policy[(size_t)g * 64 + lane];
    R.val = E.nn_value[g];
    R.nn = 0u;
    R.path_dirty = 0u;
    path_load_rest(E, R, (size_t)g, lane);
    {
        const uint32_t phase = G32(R, GW(phase));
        if (phase == RAZ_PHASE_DONE || phase == RAZ_PHASE_IDLE || G32(R, GW(error))) return;  // nn_active is already 0
    }
    if (RAZ_PROF_ON(E) && lane == 0) E.prof[(size_t)g * 8 + 5] += 1;
    const int inner_max = ((E.cfg.reserved >> 12) & 0xf) ? (int)((E.cfg.reserved >> 12) & 0xf) : kInnerMax;
    for (int it = 0; it < inner_max; ++it) {
        uint32_t phase = G32(R, GW(phase));
        if (phase == RAZ_PHASE_DONE || phase == RAZ_PHASE_IDLE) break;
        if (G32(R, GW(error))) break;
        unsigned long long t0 = prof_now();
        if (G32(R, GW(leaf_kind)) != RAZ_LEAF_NONE) backup_leaf<false>(E, R, g, G32(R, GW(player)) - 1, lane, lds64);
        prof_add(E, g, 0, t0, lane);
        t0 = prof_now();
        // controller: loop because a decided move may immediately need another decision
        // (turn-0 bypass) before a search with simulations starts
        for (int guard = 0; guard < 8; ++guard) {
            phase = G32(R, GW(phase));
            if (phase == RAZ_PHASE_NEW_MOVE) {
                begin_move<SOLVER>(E, R, g, lane, slds_p);
                continue;
            }
            if (phase == RAZ_PHASE_SEARCH && (int32_t)G32(R, GW(sims_left)) <= 0) {
                decide_move(E, R, g, lane);
                continue;
            }
            break;
        }
        prof_add(E, g, 1, t0, lane);
        phase = G32(R, GW(phase));
        if (phase != RAZ_PHASE_SEARCH || (int32_t)G32(R, GW(sims_left)) <= 0 || G32(R, GW(error))) break;
        t0 = prof_now();
        select_leaf<SOLVER, false>(E, R, g, lane, slds_p, g, 0u, 0, false);
        prof_add(E, g, 2, t0, lane);
        const uint32_t lk = G32(R, GW(leaf_kind));
        if (lk != RAZ_LEAF_TERMINAL && lk != RAZ_LEAF_SOLVED) break;  // needs the net
    }
    // write the game back: one coalesced store (+ the path when a descent ran)
    gw[lane] = R.cw;
    if (R.path_dirty) path_store(E, R, (size_t)g, lane);
    if (lane == 0) E.nn_active[g] = (uint8_t)R.nn;
}

// ------------------------------------------------------------------ parallel_search_num > 1: k_tree_par
// The reference runs simulation_num_per_move coroutines under asyncio.Semaphore(parallel_search_num)
// beside a prediction_worker that turns the queued leaves into one api.predict call
// (agent/player.py:189-215, 329-355).  raz-sched-v1 (oracle/orc_mcts.c search_moves)
// is that event loop in exact virtual time; one ROUND of it is
//   B   the batch came back: the simulations waiting for the net finish their expansion and return up
//       their paths, in the order their leaves were queued;
//   C   freed semaphore slots are taken by the next simulations, each running until it blocks;
//   D   the simulations sleeping on now_expanding whose key was expanded in B go on, in sleep order;
//   C'  slots freed in D are refilled.
// A game keeps parallel_search_num simulation SLOTS (block = raz_game layout, only the leaf_* / depth
// lanes meaningful; path and leaf-exchange rows indexed g * K + slot); lane j of three VGPRs holds
// slot j's state, order number and node slept on.  One launch = at most one round; a fill that runs
// out of its per-launch budget (games whose simulations all end on finished positions would otherwise
// be the launch's stragglers) continues at the next launch WITHOUT an intervening B, so the result
// does not depend on the budget.
#define LB(f) (1ULL << GW(f))
constexpr unsigned long long kSimLanes =
    LB(leaf_b) | (LB(leaf_b) << 1) | LB(leaf_w) | (LB(leaf_w) << 1) | LB(leaf_legal) | (LB(leaf_legal) << 1) |
    LB(leaf_kind) | LB(leaf_sym) | LB(leaf_np) | LB(depth) | LB(leaf_action) | LB(leaf_node) | LB(leaf_slot) |
    LB(leaf_tag) | LB(leaf_mirror) | LB(leaf_term_v);
#undef LB

struct Slots {
    uint32_t st, sq, pk;  // lane j = slot j: RAZ_SIM_*, order number, node slept on
};

__device__ __forceinline__ void slot_load(const raz_engine_dev& E, Regs& R, uint32_t g, uint32_t j, int lane, bool with_net) {
    const size_t gi = (size_t)g * E.K + j;
    const uint32_t v = E.sim[gi * 64 + lane];
    path_load(E, R, gi, lane, true);
    if (with_net) {
        R.pol_raw = E.nn_policy[gi * 64 + lane];
        R.val = E.nn_value[gi];
    }
    R.cw = ((kSimLanes >> lane) & 1ULL) ? v : R.cw;
    path_load_rest(E, R, gi, lane);
}
__device__ __forceinline__ void slot_store(const raz_engine_dev& E, const Regs& R, uint32_t g, uint32_t j, int lane) {
    const size_t gi = (size_t)g * E.K + j;
    if ((kSimLanes >> lane) & 1ULL) E.sim[gi * 64 + lane] = R.cw;
    path_store(E, R, gi, lane);
}
// the slot of `mask` with the smallest order number (K <= 16: a scalar scan)
__device__ __forceinline__ int pick_min_seq(uint32_t sq, unsigned long long mask) {
    int best = -1;
    uint32_t bs = 0;
    for (unsigned long long m = mask; m; m &= m - 1) {
        const int j = __ffsll((long long)m) - 1;
        const uint32_t q = lane_u32(sq, j);
        if (best < 0 || q < bs) {
            bs = q;
            best = j;
        }
    }
    return best;
}

// The kernel is ONE loop with a single call site each for the controller, the descent and the return
// path (the three large inlined bodies): duplicating them per phase doubled the code to 66 KB, more
// than the instruction cache two CUs share.  Each iteration picks the next operation of the round -
// resume the oldest queued leaf (B), start a simulation in a free slot (C, C'), wake the next sleeper
// (D) - then runs at most one slot load, one descent, one return.
template <bool SOLVER>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_tree_par(raz_engine_dev E, uint32_t g0, uint32_t count) {
    if (blockIdx.x >= count) return;
    __shared__ float lds64[64];
    __shared__ SolverLDS slds_store;
    SolverLDS* slds_p = SOLVER ? &slds_store : nullptr;
    const uint32_t g = g0 + blockIdx.x;
    const int lane = threadIdx.x;
    if (g >= E.B) return;
    const uint32_t K = E.K;
    const unsigned long long kmask = (1ULL << K) - 1ULL;  // K <= 16
    uint32_t* gw = (uint32_t*)(E.game + g);
    Regs R;
    R.cw = gw[lane];
    R.pnode = R.pmirror = R.pact = 0u;
    R.pol_raw = 0.0f;
    R.val = 0.0f;
    R.nn = 0u;
    R.path_dirty = 0u;
    Slots T;
    T.st = T.sq = T.pk = 0u;
    uint32_t* myblk = E.sim + ((size_t)g * K + (uint32_t)(lane < (int)K ? lane : 0)) * 64;
    if (lane < (int)K) {
        T.st = myblk[GW(sim_state)];
        T.sq = myblk[GW(sim_seq)];
        T.pk = myblk[GW(sim_parked)];
    }
    uint32_t nnmask = 0u;
    {
        const uint32_t phase = G32(R, GW(phase));
        if (phase == RAZ_PHASE_DONE || phase == RAZ_PHASE_IDLE || G32(R, GW(error))) {
            if (lane < (int)K) E.nn_active[(size_t)g * K + lane] = 0;
            return;
        }
    }
    if (RAZ_PROF_ON(E) && lane == 0) E.prof[(size_t)g * 8 + 5] += 1;
    int budget = (int)K + (((E.cfg.reserved >> 12) & 0xf) ? (int)((E.cfg.reserved >> 12) & 0xf) : kInnerMax);
    constexpr uint32_t kStageB = 0u, kStageC = 1u, kStageC2 = 2u, kStageD = 4u;  // D never outlives a launch
    uint32_t stage = G32(R, GW(par_stage));
    unsigned long long dmask = 0ULL;  // sleepers still to poll in D
    for (;;) {
        if (G32(R, GW(error))) break;
        // ---- the next operation of the round
        int j = -1;
        bool resume = false, wake = false;
        if (stage == kStageB) {
            const unsigned long long m = __ballot(T.st == RAZ_SIM_WAIT_NET) & kmask;
            if (!m) {
                stage = kStageC;
                continue;
            }
            j = pick_min_seq(T.sq, m);
            resume = true;
        } else if (stage == kStageD) {
            if (!dmask) {
                stage = kStageC2;
                continue;
            }
            j = pick_min_seq(T.sq, dmask);
            dmask &= ~(1ULL << j);
            wake = true;
        } else {  // C / C': the per-move controller, then a new simulation into a free slot
            for (int guard = 0; guard < 8; ++guard) {
                const uint32_t phase = G32(R, GW(phase));
                if (phase == RAZ_PHASE_NEW_MOVE) {
                    begin_move<SOLVER>(E, R, g, lane, slds_p);
                    continue;
                }
                if (phase == RAZ_PHASE_SEARCH && (int32_t)G32(R, GW(sims_left)) <= 0) {  // every simulation has returned
                    decide_move(E, R, g, lane);
                    continue;
                }
                break;
            }
            const unsigned long long busy = __ballot(T.st != RAZ_SIM_FREE) & kmask;
            const int inflight = __popcll(busy);
            const int to_start = (int32_t)G32(R, GW(sims_left)) - inflight;
            if (G32(R, GW(phase)) != RAZ_PHASE_SEARCH || G32(R, GW(error)) || inflight >= (int)K || to_start <= 0) {
                if (stage == kStageC2) {
                    stage = kStageB;  // the round is complete: the next launch starts with B
                    break;
                }
                // D: sleepers whose key is still in now_expanding sleep on (their nodes' tags are read in one go)
                const uint32_t pl = G32(R, GW(player)) - 1;
                const bool sl = lane < (int)K && T.st == RAZ_SIM_WAIT_EXPAND;
                uint32_t tg = 0u;
                if (sl) tg = node_hdr(node_ptr(E, g, T.pk))->tag;
                dmask = __ballot(sl && !((tg >> (6 + pl)) & 1u)) & kmask;
                stage = kStageD;
                continue;
            }
            if (budget <= 0) break;  // the fill goes on at the next launch, without a B in between
            --budget;
            j = __ffsll((long long)(~busy & kmask)) - 1;
        }
        // ---- at most one slot load, one descent, one return
        bool back = resume;
        if (resume || wake) slot_load(E, R, g, (uint32_t)j, lane, resume);
        if (!resume) {
            const unsigned long long t0 = prof_now();
            select_leaf<SOLVER, true>(E, R, g, lane, slds_p, g * K + (uint32_t)j,
                                      wake ? lane_u32(T.pk, j) : G32(R, GW(root_node)), wake ? (int)G32(R, GW(depth)) : 0, wake);
            prof_add(E, g, 2, t0, lane);
            const uint32_t kind = G32(R, GW(leaf_kind));
            if (kind == RAZ_LEAF_TERMINAL || kind == RAZ_LEAF_SOLVED) {
                back = true;  // ended on a finished game / a solved position: returns up its path at once
            } else if (kind == RAZ_LEAF_EXPAND || kind == RAZ_LEAF_PARKED) {
                if (kind == RAZ_LEAF_EXPAND || !wake) {  // a sleeper that goes back to sleep keeps its place
                    const uint32_t seq = G32(R, GW(par_seq_next));
                    S32(R, GW(par_seq_next), seq + 1);
                    T.sq = writelane_r(T.sq, seq, j, lane);
                }
                if (kind == RAZ_LEAF_EXPAND) {
                    T.st = writelane_r(T.st, RAZ_SIM_WAIT_NET, j, lane);
                    nnmask |= 1u << j;
                } else {
                    T.st = writelane_r(T.st, RAZ_SIM_WAIT_EXPAND, j, lane);
                    T.pk = writelane_r(T.pk, G32(R, GW(sim_parked)), j, lane);
                }
                slot_store(E, R, g, (uint32_t)j, lane);
                S32(R, GW(leaf_kind), RAZ_LEAF_NONE);
            }
        }
        if (back) {
            const unsigned long long t0 = prof_now();
            backup_leaf<true>(E, R, g, G32(R, GW(player)) - 1, lane, lds64);
            T.st = writelane_r(T.st, RAZ_SIM_FREE, j, lane);
            prof_add(E, g, 0, t0, lane);
        }
    }
    S32(R, GW(par_stage), stage == kStageD ? kStageC2 : stage);
    gw[lane] = R.cw;
    if (lane < (int)K) {
        myblk[GW(sim_state)] = T.st;
        myblk[GW(sim_seq)] = T.sq;
        myblk[GW(sim_parked)] = T.pk;
        E.nn_active[(size_t)g * K + lane] = (uint8_t)((nnmask >> lane) & 1u);
    }
}

// Reduce the per-game statistics into counters[0..6] (one block).  Per-game words instead of
// global atomics: 4096 waves hitting one address cost ~90 us per launch (one word saturates at
// ~88 atomics/us on this chip).
__global__ __launch_bounds__(256) void k_stats(raz_engine_dev E) {
    __shared__ unsigned long long sh[8][256];
    unsigned long long fin = 0, sims = 0, err = 0, leaves = 0, sel = 0, maxpool = 0, idle = 0, maxbytes = 0;
    for (uint32_t g = threadIdx.x; g < E.B; g += 256) {
        const raz_game& G = E.game[g];
        const unsigned long long pu = G.status == 0 ? G.node_count : 0, pb = G.status == 0 ? 8ULL * G.pool_used : 0;
        maxpool = pu > maxpool ? pu : maxpool;
        maxbytes = pb > maxbytes ? pb : maxbytes;
        fin += G.status != 0 ? 1 : 0;
        idle += (G.phase == RAZ_PHASE_IDLE || G.phase == RAZ_PHASE_DONE) ? 1 : 0;
        sims += G.sims;
        err |= G.error;
        leaves += G.leaves;
        sel += G.selections;
    }
    sh[0][threadIdx.x] = fin; sh[1][threadIdx.x] = sims; sh[2][threadIdx.x] = err;
    sh[3][threadIdx.x] = leaves; sh[4][threadIdx.x] = sel; sh[5][threadIdx.x] = maxpool; sh[6][threadIdx.x] = idle;
    sh[7][threadIdx.x] = maxbytes;
    __syncthreads();
    if (threadIdx.x < 8) {
        unsigned long long a = 0;
        for (int i = 0; i < 256; ++i) {
            const unsigned long long x = sh[threadIdx.x][i];
            a = (threadIdx.x == 2) ? (a | x) : ((threadIdx.x == 5 || threadIdx.x == 7) ? (x > a ? x : a) : a + x);
        }
        // + the games raz_engine_harvest took out of their slots since raz_engine_start: [8] finished, [9] sims, [10] leaves, [11] selections
        if (threadIdx.x == 0) a += E.counters[8];
        if (threadIdx.x == 1) a += E.counters[9];
        if (threadIdx.x == 3) a += E.counters[10];
        if (threadIdx.x == 4) a += E.counters[11];
        E.counters[threadIdx.x == 7 ? 16 : threadIdx.x] = a;   // ([8..15] belong to the harvest and the record extent)
    }
}

__global__ void k_start(raz_engine_dev E, uint32_t first_game_id, const uint32_t* sims_per_move,
                        uint32_t n_active) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= E.B) return;
    const bool act = g < n_active;
    raz_game G;
    memset(&G, 0, sizeof G);
    G.root_black = RAZ_INIT_BLACK;
    G.root_white = RAZ_INIT_WHITE;
    G.player = RAZ_PLAYER_BLACK;
    G.phase = act ? RAZ_PHASE_NEW_MOVE : RAZ_PHASE_IDLE;
    G.game_id = first_game_id + g;
    double d0, d1;
    raz_rng_pair(E.cfg.seed, first_game_id + g, RAZ_RNG_GAME, 0, 0, 0, d0, d1);
    G.enable_resign = E.cfg.disable_resignation_rate <= d0 ? 1 : 0;  // worker/self_play.py:144
    G.sims_per_move = sims_per_move[g];
    G.leaf_kind = RAZ_LEAF_NONE;
    G.root_node = RAZ_NO_NODE;
    G.leaf_node = RAZ_NO_NODE;
    G.leaf_mirror = RAZ_NO_NODE;
    G.pool_used = 1;   // (8-byte units; offset 0 is never a node, so that a link is never 0)
    E.game[g] = G;
    if (!E.par) E.nn_active[g] = 0;   // (slot kernel: cleared by raz_engine_start, one flag per slot)
}

// The next game of every slot ON THE SLOT'S TREE: what SelfPlayWorker.start does when it keeps `mtcs_info`
// for reset_mtcs_info_per_game games (worker/self_play.py:109-111,132-134).  Board, records, random-stream
// counters and statistics start afresh (global game id first_game_id + g); nodes, table and pool stay.
__global__ void k_next_game(raz_engine_dev E, uint32_t first_game_id, const uint32_t* sims_per_move, uint32_t n_active) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= E.B) return;
    const bool act = g < n_active;
    raz_game G;
    memset(&G, 0, sizeof G);
    G.root_black = RAZ_INIT_BLACK;
    G.root_white = RAZ_INIT_WHITE;
    G.player = RAZ_PLAYER_BLACK;
    G.phase = act ? RAZ_PHASE_NEW_MOVE : RAZ_PHASE_IDLE;
    G.game_id = first_game_id + g;
    double d0, d1;
    raz_rng_pair(E.cfg.seed, first_game_id + g, RAZ_RNG_GAME, 0, 0, 0, d0, d1);
    G.enable_resign = E.cfg.disable_resignation_rate <= d0 ? 1 : 0;  // worker/self_play.py:144
    G.sims_per_move = sims_per_move[g];
    G.leaf_kind = RAZ_LEAF_NONE;
    G.root_node = RAZ_NO_NODE;
    G.leaf_node = RAZ_NO_NODE;
    G.leaf_mirror = RAZ_NO_NODE;
    G.pool_used = E.game[g].pool_used;
    G.node_count = E.game[g].node_count;
    G.error = E.game[g].error;
    E.game[g] = G;
    if (!E.par) E.nn_active[g] = 0;
}
// The game's two new ReversiPlayers take expanded = set(var_p.keys()) (agent/player.py:47) and an empty
// now_expanding: every node that holds a prior counts as expanded for both.  grid (B, y).
__global__ __launch_bounds__(64) void k_adopt_all(raz_engine_dev E) {
    const uint32_t g = blockIdx.x;
    const int lane = threadIdx.x;
    const uint32_t used = E.game[g].node_count;
    for (uint32_t i = blockIdx.y; i < used; i += gridDim.y) {
        const uint32_t link = E.node_dir[(size_t)g * E.C + i];
        unsigned char* p = node_ptr(E, g, link);
        const int L = link_L(link);
        const bool has_p = __ballot(lane < L && node_P(p, L)[lane] != 0.0f) != 0ULL;
        raz_node_hdr* h = node_hdr(p);
        if (lane == 0) {
            uint32_t t = h->tag & ~0xC0u;
            if (has_p || ((t >> 4) & 3u)) t |= 0x30u;
            h->tag = t;
        }
    }
}

// ------------------------------------------------------------------ node pruning
// The disc count only grows, so once the real game has D discs every node whose position has fewer
// is unreachable (SURVEY.md §7 hard part 5).  k_gc compacts the pool of every game whose node count is at
// least `threshold`: kept nodes slide down in creation order (so relative order, and with it nothing
// observable, changes), every link - child, mirror, root, in-flight paths and leaves - is renumbered and the
// hash table is rebuilt.  Nodes are variable-size, so the walk goes through the game's node directory
// (creation index -> link); a link is translated through its target's header: link -> header.index ->
// remap[index] = the link after compaction.  One 256-thread workgroup per game; launched between simulation steps.
__device__ __forceinline__ uint32_t gc_translate(const raz_engine_dev& E, uint32_t g, const uint32_t* remap, uint32_t link) {
    return remap[node_hdr(node_ptr(E, g, link))->index];   // (the target still lies at its old place: nothing has moved yet)
}

__global__ __launch_bounds__(256) void k_gc(raz_engine_dev E, uint32_t threshold) {
    const uint32_t g = blockIdx.x;
    if (g >= E.B) return;
    raz_game& G = E.game[g];
    const uint32_t count = G.node_count;
    if (G.status != 0 || count < threshold) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    __shared__ uint32_t s_cnt[256], s_size[256];
    __shared__ uint32_t s_base_cnt, s_base_units;
    uint32_t* remap = E.gc_remap + (size_t)g * E.C;
    uint32_t* dir = E.node_dir + (size_t)g * E.C;
    const int dmin = bb_popcount(G.root_black) + bb_popcount(G.root_white);
    // pass 1: keep flags -> new indices and new offsets (blocked scans, 256 nodes per round)
    if (tid == 0) {
        s_base_cnt = 0;
        s_base_units = 1;   // offset 0 is never a node
    }
    __syncthreads();
    for (uint32_t i0 = 0; i0 < count; i0 += 256) {
        const uint32_t i = i0 + tid;
        uint32_t keep = 0, units = 0, link = 0;
        raz_node_hdr* h = nullptr;
        if (i < count) {
            link = dir[i];
            h = node_hdr(node_ptr(E, g, link));
            keep = (bb_popcount(h->black) + bb_popcount(h->white)) >= dmin ? 1u : 0u;
            units = keep ? node_units(link_L(link)) : 0u;
        }
        s_cnt[tid] = keep;
        s_size[tid] = units;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {  // Hillis-Steele inclusive scans
            const uint32_t v = tid >= off ? s_cnt[tid - off] : 0, u = tid >= off ? s_size[tid - off] : 0;
            __syncthreads();
            s_cnt[tid] += v;
            s_size[tid] += u;
            __syncthreads();
        }
        const uint32_t bc = s_base_cnt, bu = s_base_units;
        if (i < count) {
            remap[i] = keep ? link_make(bu + s_size[tid] - units, link_L(link)) : RAZ_NO_NODE;
            if (keep) h->gc_index = bc + s_cnt[tid] - 1;
        }
        __syncthreads();
        if (tid == 255) {
            s_base_cnt = bc + s_cnt[255];
            s_base_units = bu + s_size[255];
        }
        __syncthreads();
    }
    const uint32_t kept = s_base_cnt, kept_units = s_base_units;
    __threadfence_block();
    // pass 2: renumber the links held by the kept nodes (targets are still at their old places)
    for (uint32_t i = wv; i < count; i += 4) {
        if (remap[i] == RAZ_NO_NODE) continue;
        const uint32_t link = dir[i];
        unsigned char* p = node_ptr(E, g, link);
        const int L = link_L(link);
        if (lane < L) {
            const uint32_t c = node_child(p, L)[lane];
            if (c && !(c & 0x80000000u)) {
                const uint32_t r = gc_translate(E, g, remap, c);
                node_child(p, L)[lane] = r == RAZ_NO_NODE ? 0u : r;
            }
        }
        if (lane == 0) {
            raz_node_hdr* h = node_hdr(p);
            const uint32_t m = h->mirror;
            if (m != RAZ_NO_NODE) h->mirror = gc_translate(E, g, remap, m);
        }
    }
    // in-flight simulation state: of the game block (k_tree) or of every busy simulation slot (k_tree_par)
    const uint32_t nslots = E.par ? E.K : 1u;
    for (uint32_t j = 0; j < nslots; ++j) {
        uint32_t* blk = E.par ? E.sim + ((size_t)g * E.K + j) * 64 : (uint32_t*)&G;
        const size_t pi = E.par ? (size_t)g * E.K + j : (size_t)g;
        if (E.par && blk[GW(sim_state)] == RAZ_SIM_FREE) continue;
        if (tid < 64) {
            const uint32_t pn = E.path_node[pi * 64 + tid], pm = E.path_mirror[pi * 64 + tid];
            const int depth = (int)blk[GW(depth)];
            if (tid < depth) {
                E.path_node[pi * 64 + tid] = gc_translate(E, g, remap, pn);
                if (pm != RAZ_NO_NODE) E.path_mirror[pi * 64 + tid] = gc_translate(E, g, remap, pm);
            }
        }
        if (tid == 0) {
            const uint32_t ln = blk[GW(leaf_node)], lm = blk[GW(leaf_mirror)], lk = blk[GW(leaf_kind)];
            if (lk == RAZ_LEAF_EXPAND || lk == RAZ_LEAF_SOLVED) {
                if (ln != RAZ_NO_NODE) blk[GW(leaf_node)] = gc_translate(E, g, remap, ln);
                if (lm != RAZ_NO_NODE) blk[GW(leaf_mirror)] = gc_translate(E, g, remap, lm);
                blk[GW(leaf_slot)] = 0xfffffffeu;  // the slot found by select is gone: backup probes again
            }
            if (E.par && blk[GW(sim_state)] == RAZ_SIM_WAIT_EXPAND) blk[GW(sim_parked)] = gc_translate(E, g, remap, blk[GW(sim_parked)]);
        }
    }
    if (tid == 0) {
        const uint32_t rn = G.root_node;
        if (rn != RAZ_NO_NODE) G.root_node = gc_translate(E, g, remap, rn);
    }
    __syncthreads();
    __threadfence_block();
    // pass 3: slide the kept nodes down in creation order, four per round (read all, barrier, write all: a destination
    // never lies above its source and the sources of later rounds lie above this round's, so nothing unread is overwritten)
    for (uint32_t i0 = 0; i0 < count; i0 += 4) {
        const uint32_t i = i0 + wv;
        const uint32_t dst = i < count ? remap[i] : RAZ_NO_NODE;
        const uint32_t src = i < count ? dir[i] : 0u;
        const bool live = dst != RAZ_NO_NODE;
        const uint32_t dwords = live ? 2u * node_units(link_L(src)) : 0u;   // <= 176
        uint32_t a = 0, b = 0, c = 0;
        if (live) {
            const uint32_t* sp = (const uint32_t*)node_ptr(E, g, src);
            if ((uint32_t)lane < dwords) a = sp[lane];
            if ((uint32_t)lane + 64 < dwords) b = sp[lane + 64];
            if ((uint32_t)lane + 128 < dwords) c = sp[lane + 128];
        }
        __syncthreads();
        if (live) {
            uint32_t* dp = (uint32_t*)node_ptr(E, g, dst);
            if ((uint32_t)lane < dwords) dp[lane] = a;
            if ((uint32_t)lane + 64 < dwords) dp[lane + 64] = b;
            if ((uint32_t)lane + 128 < dwords) dp[lane + 128] = c;
        }
        __syncthreads();
        if (live && lane == 0) {   // the node now carries its new index; the directory entry of that index is free to take
            raz_node_hdr* h = node_hdr(node_ptr(E, g, dst));
            const uint32_t ni = h->gc_index;
            h->index = ni;
            dir[ni] = dst;
        }
        __syncthreads();
    }
    raz_slot* tab = E.table + (size_t)g * E.H;
    for (uint32_t sidx = tid; sidx < E.H; sidx += 256) tab[sidx].idx_tag = 0;
    if (tid == 0) {
        G.pool_used = kept_units;
        G.node_count = kept;
    }
    __syncthreads();
    __threadfence_block();
    // pass 4: rebuild the table (parallel insertion, one thread per kept node)
    const uint32_t mask = E.H - 1;
    for (uint32_t n = tid; n < kept; n += 256) {
        const uint32_t link = dir[n];
        const raz_node_hdr* h = node_hdr(node_ptr(E, g, link));
        const raz_bb kb = h->black, kw = h->white;
        const uint32_t tagkey = h->tag & RAZ_SLOT_KEYMASK;
        uint32_t si = key_hash(kb, kw, tagkey) & mask;
        const uint32_t val = RAZ_SLOT_USED | tagkey;
        for (;;) {
            if (atomicCAS(&tab[si].idx_tag, 0u, val) == 0u) {
                tab[si].black = kb;
                tab[si].white = kw;
                tab[si].link = link;
                break;
            }
            si = (si + 1) & mask;
        }
    }
}

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// parallel_search_num (config.py:142): simulation slots per game; 0 means 1.
inline size_t slots_of(const raz_engine_config& cfg) { return cfg.parallel_search_num ? cfg.parallel_search_num : 1; }
// k_tree_par drives the games when more than one simulation is in flight (or when reserved bit 3 asks for it)
inline bool uses_slot_kernel(const raz_engine_config& cfg) { return slots_of(cfg) > 1 || (cfg.reserved & 8u); }

// Bytes of one game's node pool: the caller's figure (rounded up to 8), or nodes_per_game average-size nodes plus room
// for 64 nodes of the largest size (small pools: a burst of wide positions must not trip the byte limit first).
inline unsigned long long pool_bytes_of(const raz_engine_config& cfg) {
    const unsigned long long b = cfg.pool_bytes_per_game ? cfg.pool_bytes_per_game
                                                         : (unsigned long long)cfg.nodes_per_game * RAZ_NODE_DEFAULT_BYTES + 64ULL * RAZ_NODE_MAX_BYTES;
    return (b + 7ULL) & ~7ULL;
}

// Carve the workspace; with base == nullptr only the total size is computed.
size_t carve(const raz_engine_config& cfg, unsigned char* base, raz_engine_dev* E) {
    size_t off = 0;
    const size_t B = cfg.n_games, C = cfg.nodes_per_game, H = cfg.table_slots, MP = cfg.max_plies;
    auto take = [&](size_t bytes) -> unsigned char* {
        unsigned char* p = base ? base + off : nullptr;
        off = align_up(off + bytes, 256);
        return p;
    };
    raz_engine_dev d;
    memset(&d, 0, sizeof d);
    d.cfg = cfg;
    d.B = (uint32_t)B; d.C = (uint32_t)C; d.H = (uint32_t)H; d.max_plies = (uint32_t)MP;
    d.pool_bytes = pool_bytes_of(cfg);
    const size_t K = slots_of(cfg);
    d.K = (uint32_t)K;
    d.par = uses_slot_kernel(cfg) ? 1u : 0u;
    d.game = (raz_game*)take(B * sizeof(raz_game));
    d.sim = d.par ? (uint32_t*)take(B * K * sizeof(raz_game)) : nullptr;
    const size_t BK = B * K;   // one leaf-exchange row and one path per simulation slot
    d.nn_active = take(BK);
    d.nn_own = (unsigned long long*)take(BK * 8); d.nn_enemy = (unsigned long long*)take(BK * 8);
    d.nn_policy = (float*)take(BK * 64 * 4); d.nn_value = (float*)take(BK * 4);
    d.path_node = (uint32_t*)take(BK * 64 * 4); d.path_mirror = (uint32_t*)take(BK * 64 * 4); d.path_act = (uint16_t*)take(BK * 64 * 2);
    d.table = (raz_slot*)take(B * H * sizeof(raz_slot));
    d.nodes = take(B * (size_t)d.pool_bytes);
    d.node_dir = (uint32_t*)take(B * C * 4);
    d.rec = (raz_ply_header*)take(B * MP * sizeof(raz_ply_header));
    d.rec_n = (uint32_t*)take(B * MP * 64 * 4);
    d.rec_w = cfg.record_root_w ? (double*)take(B * MP * 64 * 8) : nullptr;
    d.M = cfg.solver_memo_slots;
    d.memo = (raz_slot*)take(B * (size_t)cfg.solver_memo_slots * sizeof(raz_slot));
    d.gc_remap = (uint32_t*)take(B * C * 4);
    d.counters = (unsigned long long*)take(32 * 8);
    d.node_out = take(RAZ_NODE_OUT_BYTES + 64);
    d.prof = (unsigned long long*)take(B * 8 * 8);
    if (E) *E = d;
    return off;
}

int validate(const raz_engine_config* cfg) {
    if (!cfg) return raz_fail(RAZ_EINVAL, "raz_engine: NULL config");
    if (cfg->n_games == 0 || cfg->nodes_per_game == 0) return raz_fail(RAZ_EINVAL, "raz_engine: n_games and nodes_per_game must be > 0");
    if (cfg->table_slots < 2 * (size_t)cfg->nodes_per_game || (cfg->table_slots & (cfg->table_slots - 1)) || cfg->table_slots < RAZ_PROBE)
        return raz_fail(RAZ_EINVAL, "raz_engine: table_slots must be a power of two >= 2*nodes_per_game");
    if (cfg->max_plies < 64) return raz_fail(RAZ_EINVAL, "raz_engine: max_plies must be >= 64");
    if (pool_bytes_of(*cfg) < 2 * RAZ_NODE_MAX_BYTES + 8 || pool_bytes_of(*cfg) > 8ULL * RAZ_LINK_MAX_UNITS)
        return raz_fail(RAZ_EINVAL, "raz_engine: pool_bytes_per_game must be between 1416 bytes and 256 MB (a link holds a 25-bit offset in 8-byte units)");
    if (!(cfg->dirichlet_alpha > 0.0) || !(cfg->dirichlet_alpha < 1e6))
        return raz_fail(RAZ_EINVAL, "raz_engine: dirichlet_alpha must be a positive finite number (all shipped configs use 0.5)");
    if (cfg->thinking_loop < 1) return raz_fail(RAZ_EINVAL, "raz_engine: thinking_loop must be >= 1");
    if (cfg->parallel_search_num > 16)
        return raz_fail(RAZ_EINVAL, "raz_engine: parallel_search_num must be <= 16 (prediction_queue_size, config.py:141: the reference's queue would block beyond it)");
    if ((cfg->use_solver_turn && cfg->use_solver_turn < 46) || (cfg->use_solver_turn_in_simulation && cfg->use_solver_turn_in_simulation < 46))
        return raz_fail(RAZ_EINVAL, "raz_engine: use_solver_turn(_in_simulation) must be 0 or >= 46 (<= 14 empties; the reference relies on a 30 s timeout below that)");
    if ((cfg->use_solver_turn || cfg->use_solver_turn_in_simulation) &&
        (cfg->solver_memo_slots < 1024 || (cfg->solver_memo_slots & (cfg->solver_memo_slots - 1))))
        return raz_fail(RAZ_EINVAL, "raz_engine: solver_memo_slots must be a power of two >= 1024 when the solver is on");
    if (cfg->share_mtcs_info && !cfg->mirror_updates)
        return raz_fail(RAZ_EINVAL, "raz_engine: share_mtcs_info=1 requires mirror_updates=1 (player.py:279-280)");
    return RAZ_OK;
}

}  // namespace

constexpr int kMaxParts = 8;

// raz_leaf_cache.hip / raz_net.hip
size_t raz_leaf_cache_layout(uint32_t log2_entries, size_t rows, unsigned char* base, raz_leaf_cache_dev* out);
int raz_leaf_cache_clear(const raz_leaf_cache_dev& c, size_t rows, hipStream_t s);
int raz_leaf_cache_before(const raz_leaf_cache_dev& c, const raz_engine_dev& d, uint32_t p0, uint32_t pn, uint32_t part, uint32_t step,
                          hipStream_t s);
int raz_leaf_cache_after(const raz_leaf_cache_dev& c, const raz_engine_dev& d, uint32_t p0, uint32_t pn, uint32_t step, hipStream_t s);
int raz_net_forward_compact(const raz_net* net, const uint64_t* own, const uint64_t* enemy, const uint8_t* active, float* policy,
                            float* value, size_t n, void* scratch, size_t scratch_bytes, hipStream_t stream, const uint32_t* list,
                            const uint32_t* n_ptr);

// The internal streams of all engines of a process, per device, created once and never destroyed.  HIP maps streams onto a
// few hardware queues round-robin in creation order: an engine that creates fresh streams after other engines have come
// and gone can land two of its slices on ONE queue, which serialises them (measured: the same workload 3.3x slower after
// an engine with a different stream history had been destroyed).  A fixed pool keeps every engine on the mapping of the
// first one.  Engines of one host thread run one at a time (include/raz.h), so sharing the streams orders nothing extra.
static hipError_t raz_pool_stream(int h, hipStream_t* out) {
    static std::mutex mu;
    static hipStream_t pool[64][kMaxParts] = {};
    int dev = 0;
    hipError_t err = hipGetDevice(&dev);
    if (err != hipSuccess) return err;
    if (dev < 0 || dev >= 64 || h < 0 || h >= kMaxParts) return hipErrorInvalidValue;
    std::lock_guard<std::mutex> lock(mu);
    for (int i = 0; i <= h; ++i)   // always in index order, so that stream i of a device is the i-th created
        if (!pool[dev][i]) {
            err = hipStreamCreateWithFlags(&pool[dev][i], hipStreamNonBlocking);
            if (err != hipSuccess) return err;
        }
    *out = pool[dev][h];
    return hipSuccess;
}

struct raz_engine {
    raz_engine_dev dev;
    raz_net net;
    void* net_scratch;
    size_t net_scratch_bytes;
    uint32_t* d_sims;  // staging for sims_per_move
    double* d_thr;     // staging for per-game resign thresholds (raz_engine_harvest)
    raz_leaf_cache_dev cache;   // cross-game evaluation cache (raz_engine_set_leaf_cache); cache.tags == nullptr: none
    uint32_t cache_step;        // stamp of the next half step
    bool started;
    // The batch is stepped as `parts` independent slices on as many streams (the caller's and
    // parts-1 internal ones): while one slice's leaves are in the net kernel (matrix pipe) another
    // slice's tree kernel (scalar / f64 VALU, latency-bound) runs beside it instead of after it.
    int parts;
    hipStream_t aux[kMaxParts];
    hipEvent_t ev_fork, ev_join[kMaxParts];
    // kGraphSteps simulation steps (parts x 2 kernels each, fork/join included) captured once as a
    // hipGraph and replayed: one host call per kGraphSteps * parts * 2 launches.
    hipGraphExec_t graph_exec;
    hipStream_t graph_stream;   // the caller stream the graph was captured on
    int graph_parts;
    bool graph_off;             // cfg.reserved bit 2 not set, or capture failed once
};

namespace {

void drop_graph(raz_engine* e) {
    if (e->graph_exec) hipGraphExecDestroy(e->graph_exec);
    e->graph_exec = nullptr;
}

struct Half {
    uint32_t g0, count;
};
inline Half half_of(const raz_engine* e, int h) {
    const uint32_t B = e->dev.B, P = (uint32_t)e->parts;
    const uint32_t per = ((B + P - 1) / P + 63) / 64 * 64;  // slice size, multiple of 64
    const uint32_t g0 = per * (uint32_t)h < B ? per * (uint32_t)h : B;
    const uint32_t g1 = g0 + per < B ? g0 + per : B;
    return Half{g0, g1 - g0};
}
inline hipStream_t stream_of(const raz_engine* e, int h, hipStream_t s) { return h == 0 ? s : e->aux[h]; }

// one simulation step of one slice on stream s; ev (nullable) = 3 events bracketing the two kernels
int launch_half_step(raz_engine* e, int h, hipStream_t s, hipEvent_t* ev) {
    const raz_engine_dev& d = e->dev;
    const Half hf = half_of(e, h);
    if (hf.count == 0) return RAZ_OK;
    if (ev) hipEventRecord(ev[0], s);
    const bool solver = d.cfg.use_solver_turn || d.cfg.use_solver_turn_in_simulation;
    if (d.par) {
        if (solver)
            hipLaunchKernelGGL(k_tree_par<true>, dim3(hf.count), dim3(64), 0, s, d, hf.g0, hf.count);
        else
            hipLaunchKernelGGL(k_tree_par<false>, dim3(hf.count), dim3(64), 0, s, d, hf.g0, hf.count);
    } else if (solver)
        hipLaunchKernelGGL(k_tree<true>, dim3(hf.count), dim3(64), 0, s, d, hf.g0, hf.count);
    else
        hipLaunchKernelGGL(k_tree<false>, dim3(hf.count), dim3(64), 0, s, d, hf.g0, hf.count);
    int rc = raz_check_launch("raz_engine_step: k_tree");
    if (rc != RAZ_OK) return rc;
    if (ev) hipEventRecord(ev[1], s);
    // the slices run concurrently: each gets its own part of the net scratch (size is linear in n);
    // a game contributes one leaf-exchange row per simulation slot
    const size_t p0 = (size_t)hf.g0 * d.K, pn = (size_t)hf.count * d.K;
    const size_t soff = raz_net_scratch_bytes(e->net.filters, e->net.value_fc, p0);
    const size_t sbytes = raz_net_scratch_bytes(e->net.filters, e->net.value_fc, pn);
    if (e->cache.tags) {
        // positions the table already holds (or that another row of this batch is evaluating) leave the batch; the rest is
        // evaluated over a compact list and its answers are added to the table
        const uint32_t step = e->cache_step++;
        rc = raz_leaf_cache_before(e->cache, d, (uint32_t)p0, (uint32_t)pn, (uint32_t)h, step, s);
        if (rc != RAZ_OK) return rc;
        rc = raz_net_forward_compact(&e->net, (const uint64_t*)d.nn_own + p0, (const uint64_t*)d.nn_enemy + p0, d.nn_active + p0,
                                     d.nn_policy + p0 * 64, d.nn_value + p0, pn,
                                     e->net_scratch ? (unsigned char*)e->net_scratch + soff : nullptr, sbytes, s,
                                     e->cache.list + p0, e->cache.n_compact + h);
        if (rc != RAZ_OK) return rc;
        rc = raz_leaf_cache_after(e->cache, d, (uint32_t)p0, (uint32_t)pn, step, s);
    } else
        rc = raz_net_forward(&e->net, (const uint64_t*)d.nn_own + p0, (const uint64_t*)d.nn_enemy + p0,
                             d.nn_active + p0, d.nn_policy + p0 * 64, d.nn_value + p0, pn,
                             e->net_scratch ? (unsigned char*)e->net_scratch + soff : nullptr, sbytes, (raz_stream_t)s);
    if (ev) hipEventRecord(ev[2], s);
    return rc;
}

int fork_aux(raz_engine* e, hipStream_t s) {
    if (e->parts == 1) return RAZ_OK;
    RAZ_HIP_TRY(hipEventRecord(e->ev_fork, s), "raz_engine_step: fork record");
    for (int h = 1; h < e->parts; ++h)
        RAZ_HIP_TRY(hipStreamWaitEvent(e->aux[h], e->ev_fork, 0), "raz_engine_step: fork wait");
    return RAZ_OK;
}
int join_aux(raz_engine* e, hipStream_t s) {
    for (int h = 1; h < e->parts; ++h) {
        RAZ_HIP_TRY(hipEventRecord(e->ev_join[h], e->aux[h]), "raz_engine_step: join record");
        RAZ_HIP_TRY(hipStreamWaitEvent(s, e->ev_join[h], 0), "raz_engine_step: join wait");
    }
    return RAZ_OK;
}

}  // namespace

extern "C" void raz_engine_destroy(raz_engine* e);

extern "C" size_t raz_engine_workspace_bytes(const raz_engine_config* cfg) {
    if (validate(cfg) != RAZ_OK) return 0;
    return carve(*cfg, nullptr, nullptr) + align_up((size_t)cfg->n_games * 4, 256) + align_up((size_t)cfg->n_games * 8, 256);
}

extern "C" int raz_engine_create(const raz_engine_config* cfg, const raz_net* net, void* d_workspace,
                                 size_t workspace_bytes, void* d_net_scratch, size_t net_scratch_bytes,
                                 raz_engine** out) {
    if (!out) return raz_fail(RAZ_EINVAL, "raz_engine_create: NULL out");
    *out = nullptr;
    int rc = validate(cfg);
    if (rc != RAZ_OK) return rc;
    if (!net || !net->d_weights) return raz_fail(RAZ_EINVAL, "raz_engine_create: net not loaded");
    if (!d_workspace || ((uintptr_t)d_workspace & 255)) return raz_fail(RAZ_EINVAL, "raz_engine_create: workspace must be 256-byte aligned");
    const size_t need = raz_engine_workspace_bytes(cfg);
    if (workspace_bytes < need) return raz_fail(RAZ_ENOMEM, "raz_engine_create: workspace too small (raz_engine_workspace_bytes)");
    if (net_scratch_bytes < raz_net_scratch_bytes(net->filters, net->value_fc, (size_t)cfg->n_games * slots_of(*cfg)))
        return raz_fail(RAZ_ENOMEM, "raz_engine_create: net scratch too small (raz_net_scratch_bytes)");
    raz_engine* e = new (std::nothrow) raz_engine;
    if (!e) return raz_fail(RAZ_ENOMEM, "raz_engine_create: host allocation failed");
    const size_t used = carve(*cfg, (unsigned char*)d_workspace, &e->dev);
    e->d_sims = (uint32_t*)((unsigned char*)d_workspace + used);
    e->d_thr = (double*)((unsigned char*)d_workspace + used + align_up((size_t)cfg->n_games * 4, 256));
    e->net = *net;
    e->net_scratch = d_net_scratch;
    e->net_scratch_bytes = net_scratch_bytes;
    e->started = false;
    memset(&e->cache, 0, sizeof e->cache);
    e->cache_step = 1;
    // reserved bit 1: single stream; bits 8..11: number of slices (default 3); bits 12..15: kInnerMax override
    int parts = (int)((cfg->reserved >> 8) & 0xf);
    if (parts == 0) parts = 3;
    if (parts > kMaxParts) parts = kMaxParts;
    if (cfg->n_games < 256 || (cfg->reserved & 2u)) parts = 1;
    e->parts = parts;
    e->graph_exec = nullptr;
    e->graph_stream = nullptr;
    e->graph_parts = 0;
    e->graph_off = (cfg->reserved & 4u) == 0;   // reserved bit 2: replay captured hipGraphs (measured slower on ROCm 7.2: off by default)
    e->ev_fork = nullptr;
    for (int h = 0; h < kMaxParts; ++h) {
        e->aux[h] = nullptr;
        e->ev_join[h] = nullptr;
    }
    if (parts > 1) {
        hipError_t err = hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming);
        for (int h = 1; h < parts && err == hipSuccess; ++h) {
            err = raz_pool_stream(h, &e->aux[h]);
            if (err == hipSuccess) err = hipEventCreateWithFlags(&e->ev_join[h], hipEventDisableTiming);
        }
        if (err != hipSuccess) {
            raz_engine_destroy(e);
            return raz_fail_hip(err, "raz_engine_create: stream/event creation");
        }
    }
    *out = e;
    return RAZ_OK;
}

// Change the number of slices/streams the batch is stepped in (1..8).  Takes effect at the next
// raz_engine_step call; the caller must have synchronised the stream.
extern "C" int raz_engine_set_parts(raz_engine* e, int parts) {
    if (!e || parts < 1 || parts > kMaxParts) return raz_fail(RAZ_EINVAL, "raz_engine_set_parts: parts must be 1..8");
    if (e->dev.B < 256) parts = 1;
    hipError_t err = hipSuccess;
    if (parts > 1 && !e->ev_fork) err = hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming);
    for (int h = 1; h < parts && err == hipSuccess; ++h) {
        if (!e->aux[h]) err = raz_pool_stream(h, &e->aux[h]);
        if (err == hipSuccess && !e->ev_join[h]) err = hipEventCreateWithFlags(&e->ev_join[h], hipEventDisableTiming);
    }
    if (err != hipSuccess) return raz_fail_hip(err, "raz_engine_set_parts");
    if (parts != e->parts) drop_graph(e);
    e->parts = parts;
    return RAZ_OK;
}

extern "C" void raz_engine_destroy(raz_engine* e) {
    if (!e) return;
    for (int h = 0; h < kMaxParts; ++h) {
        // (aux streams belong to the process-wide pool)
        if (e->ev_join[h]) hipEventDestroy(e->ev_join[h]);
    }
    if (e->ev_fork) hipEventDestroy(e->ev_fork);
    if (e->graph_exec) hipGraphExecDestroy(e->graph_exec);
    delete e;
}

extern "C" int raz_engine_start(raz_engine* e, uint32_t first_game_id, const uint32_t* sims_per_move,
                                uint32_t n_active, raz_stream_t stream) {
    if (!e || !sims_per_move) return raz_fail(RAZ_EINVAL, "raz_engine_start: NULL argument");
    if (n_active > e->dev.B) return raz_fail(RAZ_EINVAL, "raz_engine_start: n_active > n_games");
    hipStream_t s = (hipStream_t)stream;
    const raz_engine_dev& d = e->dev;
    RAZ_HIP_TRY(hipMemcpyAsync(e->d_sims, sims_per_move, (size_t)d.B * 4, hipMemcpyHostToDevice, s), "raz_engine_start: copy sims");
    RAZ_HIP_TRY(hipStreamSynchronize(s), "raz_engine_start: sync");  // host array may be transient
    RAZ_HIP_TRY(hipMemsetAsync(d.table, 0, (size_t)d.B * d.H * sizeof(raz_slot), s), "raz_engine_start: clear tables");
    if (d.M) RAZ_HIP_TRY(hipMemsetAsync(d.memo, 0, (size_t)d.B * d.M * sizeof(raz_slot), s), "raz_engine_start: clear solver memo");
    RAZ_HIP_TRY(hipMemsetAsync(d.counters, 0, 256, s), "raz_engine_start: clear counters");
    if (d.par) {
        RAZ_HIP_TRY(hipMemsetAsync(d.sim, 0, (size_t)d.B * d.K * sizeof(raz_game), s), "raz_engine_start: clear simulation slots");
        RAZ_HIP_TRY(hipMemsetAsync(d.nn_active, 0, (size_t)d.B * d.K, s), "raz_engine_start: clear leaf flags");
    }
    RAZ_HIP_TRY(hipMemsetAsync(d.prof, 0, (size_t)d.B * 64, s), "raz_engine_start: clear profile");
    hipLaunchKernelGGL(k_start, dim3((d.B + 255) / 256), dim3(256), 0, s, d, first_game_id, e->d_sims, n_active);
    int rc = raz_check_launch("raz_engine_start");
    if (rc == RAZ_OK) e->started = true;
    return rc;
}

// SelfPlayWorker.start keeps its MCTSInfo for reset_mtcs_info_per_game games (worker/self_play.py:109-111,
// 132-134): start the NEXT game of every slot on the slot's tree (share_mtcs_info only - without it the
// reference gives every game's players fresh trees, and this is raz_engine_start).
extern "C" int raz_engine_next_game(raz_engine* e, uint32_t first_game_id, const uint32_t* sims_per_move,
                                    uint32_t n_active, raz_stream_t stream) {
    if (!e || !sims_per_move) return raz_fail(RAZ_EINVAL, "raz_engine_next_game: NULL argument");
    if (!e->started) return raz_fail(RAZ_ESTATE, "raz_engine_next_game: call raz_engine_start first");
    if (!e->dev.cfg.share_mtcs_info) return raz_engine_start(e, first_game_id, sims_per_move, n_active, stream);
    if (n_active > e->dev.B) return raz_fail(RAZ_EINVAL, "raz_engine_next_game: n_active > n_games");
    hipStream_t s = (hipStream_t)stream;
    const raz_engine_dev& d = e->dev;
    RAZ_HIP_TRY(hipMemcpyAsync(e->d_sims, sims_per_move, (size_t)d.B * 4, hipMemcpyHostToDevice, s), "raz_engine_next_game: copy sims");
    RAZ_HIP_TRY(hipStreamSynchronize(s), "raz_engine_next_game: sync");  // host array may be transient
    if (d.M) RAZ_HIP_TRY(hipMemsetAsync(d.memo, 0, (size_t)d.B * d.M * sizeof(raz_slot), s), "raz_engine_next_game: clear solver memo");
    RAZ_HIP_TRY(hipMemsetAsync(d.counters, 0, 256, s), "raz_engine_next_game: clear counters");
    if (d.par) {
        RAZ_HIP_TRY(hipMemsetAsync(d.sim, 0, (size_t)d.B * d.K * sizeof(raz_game), s), "raz_engine_next_game: clear simulation slots");
        RAZ_HIP_TRY(hipMemsetAsync(d.nn_active, 0, (size_t)d.B * d.K, s), "raz_engine_next_game: clear leaf flags");
    }
    hipLaunchKernelGGL(k_adopt_all, dim3(d.B, 8), dim3(64), 0, s, d);
    int rc = raz_check_launch("raz_engine_next_game: adopt");
    if (rc != RAZ_OK) return rc;
    hipLaunchKernelGGL(k_next_game, dim3((d.B + 255) / 256), dim3(256), 0, s, d, first_game_id, e->d_sims, n_active);
    return raz_check_launch("raz_engine_next_game");
}

namespace {
constexpr uint32_t kGraphSteps = 16;

int launch_steps_direct(raz_engine* e, uint32_t n_steps, hipStream_t s) {
    int rc = fork_aux(e, s);
    for (uint32_t i = 0; i < n_steps && rc == RAZ_OK; ++i) {
        for (int h = 0; h < e->parts && rc == RAZ_OK; ++h) rc = launch_half_step(e, h, stream_of(e, h, s), nullptr);
    }
    const int rj = join_aux(e, s);
    return rc != RAZ_OK ? rc : rj;
}

// Capture kGraphSteps steps on (s, aux streams).  On any failure the engine keeps launching directly.
bool ensure_graph(raz_engine* e, hipStream_t s) {
    if (e->graph_off) return false;
    if (e->cache.tags) return false;   // the cache's step stamp is a host-side kernel argument: a replayed graph would freeze it
    if (e->graph_exec && e->graph_stream == s && e->graph_parts == e->parts) return true;
    drop_graph(e);
    hipGraph_t g = nullptr;
    hipError_t eb = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    if (eb != hipSuccess) {
        (void)hipGetLastError();
        raz_fail_hip(eb, "raz_engine_step: hipStreamBeginCapture (falling back to direct launches)");
        e->graph_off = true;
        return false;
    }
    const int rc = launch_steps_direct(e, kGraphSteps, s);
    const hipError_t ec = hipStreamEndCapture(s, &g);
    if (rc != RAZ_OK || ec != hipSuccess || !g) {
        (void)hipGetLastError();
        if (ec != hipSuccess) raz_fail_hip(ec, "raz_engine_step: hipStreamEndCapture (falling back to direct launches)");
        if (g) hipGraphDestroy(g);
        e->graph_off = true;
        return false;
    }
    const hipError_t ei = hipGraphInstantiate(&e->graph_exec, g, nullptr, nullptr, 0);
    hipGraphDestroy(g);
    if (ei != hipSuccess) {
        (void)hipGetLastError();
        raz_fail_hip(ei, "raz_engine_step: hipGraphInstantiate (falling back to direct launches)");
        e->graph_exec = nullptr;
        e->graph_off = true;
        return false;
    }
    e->graph_stream = s;
    e->graph_parts = e->parts;
    return true;
}
}  // namespace

extern "C" int raz_engine_step(raz_engine* e, uint32_t n_steps, raz_stream_t stream) {
    if (!e) return raz_fail(RAZ_EINVAL, "raz_engine_step: NULL engine");
    if (!e->started) return raz_fail(RAZ_ESTATE, "raz_engine_step: call raz_engine_start first");
    hipStream_t s = (hipStream_t)stream;
    if (n_steps < kGraphSteps || e->graph_off) return launch_steps_direct(e, n_steps, s);
    // The legacy default stream cannot be captured: run on an internal stream ordered after / before it.
    hipStream_t run = s;
    if (s == nullptr) {
        if (!e->aux[0]) {
            hipError_t err = raz_pool_stream(0, &e->aux[0]);
            if (err == hipSuccess) err = hipEventCreateWithFlags(&e->ev_join[0], hipEventDisableTiming);
            if (err == hipSuccess && !e->ev_fork) err = hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming);
            if (err != hipSuccess) return raz_fail_hip(err, "raz_engine_step: internal stream");
        }
        run = e->aux[0];
        RAZ_HIP_TRY(hipEventRecord(e->ev_join[0], s), "raz_engine_step: order after the caller stream");
        RAZ_HIP_TRY(hipStreamWaitEvent(run, e->ev_join[0], 0), "raz_engine_step: order after the caller stream");
    }
    int rc = RAZ_OK;
    if (ensure_graph(e, run)) {
        while (n_steps >= kGraphSteps) {
            hipError_t err = hipGraphLaunch(e->graph_exec, run);
            if (err != hipSuccess) return raz_fail_hip(err, "raz_engine_step: hipGraphLaunch");
            n_steps -= kGraphSteps;
        }
    }
    if (n_steps) rc = launch_steps_direct(e, n_steps, run);
    if (run != s) {
        RAZ_HIP_TRY(hipEventRecord(e->ev_join[0], run), "raz_engine_step: order before the caller stream");
        RAZ_HIP_TRY(hipStreamWaitEvent(s, e->ev_join[0], 0), "raz_engine_step: order before the caller stream");
    }
    return rc;
}

// 1 when raz_engine_step replays a captured hipGraph, 0 when it launches kernel by kernel.
extern "C" int raz_engine_uses_graph(const raz_engine* e) { return e && e->graph_exec ? 1 : 0; }

extern "C" int raz_engine_stats_sync(raz_engine* e, raz_engine_stats* out, raz_stream_t stream) {
    if (!e || !out) return raz_fail(RAZ_EINVAL, "raz_engine_stats_sync: NULL argument");
    unsigned long long c[17];
    hipLaunchKernelGGL(k_stats, dim3(1), dim3(256), 0, (hipStream_t)stream, e->dev);
    RAZ_HIP_TRY(hipMemcpyAsync(c, e->dev.counters, sizeof c, hipMemcpyDeviceToHost, (hipStream_t)stream), "raz_engine_stats_sync: copy");
    RAZ_HIP_TRY(hipStreamSynchronize((hipStream_t)stream), "raz_engine_stats_sync: sync");
    out->finished_games = c[0];
    out->total_sims = c[1];
    out->error_flags = c[2];
    out->nn_leaves = c[3];
    out->selections = c[4];
    out->max_pool_used = c[5];
    out->idle_or_done = c[6];
    out->max_pool_bytes = c[16];
    return RAZ_OK;
}

namespace {
__global__ void k_set_position(raz_engine_dev E, uint32_t g, unsigned long long black, unsigned long long white,
                               uint32_t player, uint32_t sims, uint32_t enable_resign, uint32_t one_move) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    raz_game& G = E.game[g];
    G.root_black = black;
    G.root_white = white;
    G.player = player;
    G.status = 0;
    G.phase = RAZ_PHASE_NEW_MOVE;
    G.sims_per_move = sims;
    G.sims_left = 0;
    G.loops_done = 0;
    G.move_sims = 0;
    G.leaf_kind = RAZ_LEAF_NONE;
    if (!E.par) E.nn_active[g] = 0;   // (slot kernel: no simulation is in flight between two moves of a slot)
    G.par_stage = 0;
    G.enable_resign = enable_resign;
    G.one_move = one_move;
    G.root_node = RAZ_NO_NODE;
    if (one_move) G.n_plies = 0;  // the facade reads each ply back right after it is decided
}

// The same for slots first_slot .. first_slot + n - 1 at once, positions in device arrays (one thread per slot).
__global__ void k_set_positions(raz_engine_dev E, uint32_t first_slot, uint32_t n, const unsigned long long* black,
                                const unsigned long long* white, const uint8_t* player, uint32_t sims, uint32_t enable_resign,
                                uint32_t one_move) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t g = first_slot + i;
    raz_game& G = E.game[g];
    G.root_black = black[i];
    G.root_white = white[i];
    G.player = player[i];
    G.status = 0;
    G.phase = RAZ_PHASE_NEW_MOVE;
    G.sims_per_move = sims;
    G.sims_left = 0;
    G.loops_done = 0;
    G.move_sims = 0;
    G.leaf_kind = RAZ_LEAF_NONE;
    if (!E.par) E.nn_active[g] = 0;
    G.par_stage = 0;
    G.enable_resign = enable_resign;
    G.one_move = one_move;
    G.root_node = RAZ_NO_NODE;
    if (one_move) G.n_plies = 0;
}

// var_n[key] / var_w[key] / var_p[key] of one slot: copy the node of (black, white, next_player)
// owned by `owner` to node_out (found flag in the trailing word); never creates a node.
__global__ __launch_bounds__(64) void k_read_node(raz_engine_dev E, uint32_t g, unsigned long long black,
                                                  unsigned long long white, uint32_t np, uint32_t owner) {
    const int lane = threadIdx.x;
    const Found f = table_find(E, g, black, white, np | (owner << 2), lane);
    uint32_t* flag = (uint32_t*)(E.node_out + RAZ_NODE_OUT_BYTES);
    if (lane == 0) *flag = f.found ? 1u : 0u;
    if (!f.found) return;
    unsigned char* p = node_ptr(E, g, f.node);
    const int L = link_L(f.node);
    const raz_bb legal = node_hdr(p)->legal;
    const bool on = (legal >> lane) & 1ULL;
    const int rk = rank_of(legal, lane);
    ((double*)E.node_out)[lane] = on ? node_W(p, L)[rk] : 0.0;
    ((uint32_t*)(E.node_out + 512))[lane] = on ? node_N(p, L)[rk] : 0u;
    ((float*)(E.node_out + 768))[lane] = on ? node_P(p, L)[rk] : 0.0f;
}
}  // namespace

// Put slot `slot` on the position (black, white, `player` to move) WITHOUT touching its search tree,
// random-stream counters or records, and arm a move with `sims` simulations.  one_move != 0: the slot
// idles after deciding that move (ReversiPlayer.action_with_evaluation, agent/player.py:82-134, which
// keeps var_n / var_w / var_p across calls); 0: the game continues from there.
namespace {
// ReversiPlayer.stop_thinking (agent/player.py:163-164,196-199): end the running search at the next
// step and decide with what the tree holds (no further thinking loop).
__global__ void k_stop_thinking(raz_engine_dev E, uint32_t g) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    raz_game& G = E.game[g];
    if (G.phase != RAZ_PHASE_SEARCH) return;
    int inflight = 0;   // simulations already started return first (requested_stop_thinking only stops new ones, :206-208)
    if (E.par)
        for (uint32_t j = 0; j < E.K; ++j) inflight += E.sim[((size_t)g * E.K + j) * 64 + GW(sim_state)] != RAZ_SIM_FREE;
    if (G.sims_left > inflight) G.sims_left = inflight;
    G.loops_done = (uint32_t)E.cfg.thinking_loop;
}
// A new ReversiPlayer built on a used MCTSInfo starts with expanded = set(var_p.keys())
// (agent/player.py:47): every key that holds a prior counts as expanded for player index `pl`.
__global__ __launch_bounds__(64) void k_adopt_tree(raz_engine_dev E, uint32_t g, uint32_t pl) {
    const int lane = threadIdx.x;
    const uint32_t used = E.game[g].node_count;
    for (uint32_t i = blockIdx.x; i < used; i += gridDim.x) {
        const uint32_t link = E.node_dir[(size_t)g * E.C + i];
        unsigned char* p = node_ptr(E, g, link);
        const int L = link_L(link);
        const bool has_p = __ballot(lane < L && node_P(p, L)[lane] != 0.0f) != 0ULL;
        raz_node_hdr* h = node_hdr(p);
        if (lane == 0 && (has_p || ((h->tag >> 4) & 3u))) h->tag |= 1u << (4 + pl);
    }
}
}  // namespace

extern "C" int raz_engine_stop_thinking(raz_engine* e, uint32_t slot, raz_stream_t stream) {
    if (!e) return raz_fail(RAZ_EINVAL, "raz_engine_stop_thinking: NULL engine");
    if (!e->started || slot >= e->dev.B) return raz_fail(RAZ_EINVAL, "raz_engine_stop_thinking: not started / bad slot");
    hipLaunchKernelGGL(k_stop_thinking, dim3(1), dim3(64), 0, (hipStream_t)stream, e->dev, slot);
    return raz_check_launch("raz_engine_stop_thinking");
}

extern "C" int raz_engine_adopt_tree(raz_engine* e, uint32_t slot, int player_index, raz_stream_t stream) {
    if (!e) return raz_fail(RAZ_EINVAL, "raz_engine_adopt_tree: NULL engine");
    if (!e->started || slot >= e->dev.B || player_index < 0 || player_index > 1)
        return raz_fail(RAZ_EINVAL, "raz_engine_adopt_tree: not started / bad slot / bad player index");
    hipLaunchKernelGGL(k_adopt_tree, dim3(256), dim3(64), 0, (hipStream_t)stream, e->dev, slot, (uint32_t)player_index);
    return raz_check_launch("raz_engine_adopt_tree");
}

extern "C" int raz_engine_read_node(raz_engine* e, uint32_t slot, uint64_t black, uint64_t white, int next_player,
                                    int owner, double* w64, uint32_t* n64, float* p64, int* found,
                                    raz_stream_t stream) {
    if (!e || !found) return raz_fail(RAZ_EINVAL, "raz_engine_read_node: NULL argument");
    if (!e->started) return raz_fail(RAZ_ESTATE, "raz_engine_read_node: call raz_engine_start first");
    if (slot >= e->dev.B || (next_player != 1 && next_player != 2) || owner < 0 || owner > 1)
        return raz_fail(RAZ_EINVAL, "raz_engine_read_node: bad slot / next_player / owner");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_read_node, dim3(1), dim3(64), 0, s, e->dev, slot, (unsigned long long)black,
                       (unsigned long long)white, (uint32_t)next_player, (uint32_t)owner);
    int rc = raz_check_launch("raz_engine_read_node");
    if (rc != RAZ_OK) return rc;
    unsigned char host[RAZ_NODE_OUT_BYTES + 64];
    RAZ_HIP_TRY(hipMemcpyAsync(host, e->dev.node_out, sizeof(host), hipMemcpyDeviceToHost, s), "raz_engine_read_node: copy");
    RAZ_HIP_TRY(hipStreamSynchronize(s), "raz_engine_read_node: sync");
    uint32_t flag;
    memcpy(&flag, host + RAZ_NODE_OUT_BYTES, 4);
    *found = (int)flag;
    if (flag) {
        if (w64) memcpy(w64, host, 64 * sizeof(double));
        if (n64) memcpy(n64, host + 512, 64 * sizeof(uint32_t));
        if (p64) memcpy(p64, host + 768, 64 * sizeof(float));
    } else {
        if (w64) memset(w64, 0, 64 * sizeof(double));
        if (n64) memset(n64, 0, 64 * sizeof(uint32_t));
        if (p64) memset(p64, 0, 64 * sizeof(float));
    }
    return RAZ_OK;
}

extern "C" int raz_engine_set_position(raz_engine* e, uint32_t slot, uint64_t black, uint64_t white, int player,
                                       uint32_t sims, int enable_resign, int one_move, raz_stream_t stream) {
    if (!e) return raz_fail(RAZ_EINVAL, "raz_engine_set_position: NULL engine");
    if (!e->started) return raz_fail(RAZ_ESTATE, "raz_engine_set_position: call raz_engine_start first");
    if (slot >= e->dev.B || (player != 1 && player != 2) || sims == 0)
        return raz_fail(RAZ_EINVAL, "raz_engine_set_position: bad slot / player / sims");
    hipLaunchKernelGGL(k_set_position, dim3(1), dim3(64), 0, (hipStream_t)stream, e->dev, slot,
                       (unsigned long long)black, (unsigned long long)white, (uint32_t)player, sims,
                       (uint32_t)(enable_resign != 0), (uint32_t)(one_move != 0));
    return raz_check_launch("raz_engine_set_position");
}

extern "C" int raz_engine_set_positions(raz_engine* e, uint32_t first_slot, uint32_t n, const uint64_t* d_black, const uint64_t* d_white,
                                        const uint8_t* d_player, uint32_t sims, int enable_resign, int one_move, raz_stream_t stream) {
    if (!e || !d_black || !d_white || !d_player) return raz_fail(RAZ_EINVAL, "raz_engine_set_positions: NULL argument");
    if (!e->started) return raz_fail(RAZ_ESTATE, "raz_engine_set_positions: call raz_engine_start first");
    if ((size_t)first_slot + n > e->dev.B || sims == 0) return raz_fail(RAZ_EINVAL, "raz_engine_set_positions: bad slot range / sims");
    if (n == 0) return RAZ_OK;
    hipLaunchKernelGGL(k_set_positions, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, e->dev, first_slot, n,
                       (const unsigned long long*)d_black, (const unsigned long long*)d_white, d_player, sims,
                       (uint32_t)(enable_resign != 0), (uint32_t)(one_move != 0));
    return raz_check_launch("raz_engine_set_positions");
}

// Prune unreachable nodes (positions with fewer discs than the current real position) in every
// game whose pool holds at least `threshold` nodes.  Asynchronous on `stream`; call between
// raz_engine_step calls.  Results are unaffected (kept nodes keep their relative order).
extern "C" int raz_engine_gc(raz_engine* e, uint32_t threshold, raz_stream_t stream) {
    if (!e) return raz_fail(RAZ_EINVAL, "raz_engine_gc: NULL engine");
    if (!e->started) return raz_fail(RAZ_ESTATE, "raz_engine_gc: call raz_engine_start first");
    hipLaunchKernelGGL(k_gc, dim3(e->dev.B), dim3(256), 0, (hipStream_t)stream, e->dev, threshold);
    return raz_check_launch("raz_engine_gc");
}

// raz_engine_step with HIP events around every kernel launch on `stream` (the stream the kernels
// run on): adds the elapsed milliseconds of the tree kernel and of the net kernel to *tree_ms /
// *net_ms.  Synchronises the stream.  Used by bench.py for the live roofline numbers.
extern "C" int raz_engine_step_timed(raz_engine* e, uint32_t n_steps, double* tree_ms, double* net_ms,
                                     raz_stream_t stream) {
    if (!e || !tree_ms || !net_ms) return raz_fail(RAZ_EINVAL, "raz_engine_step_timed: NULL argument");
    if (!e->started) return raz_fail(RAZ_ESTATE, "raz_engine_step_timed: call raz_engine_start first");
    hipStream_t s = (hipStream_t)stream;
    const int H = e->parts;
    std::vector<hipEvent_t> ev(3 * (size_t)n_steps * H);
    for (auto& x : ev) RAZ_HIP_TRY(hipEventCreate(&x), "raz_engine_step_timed: hipEventCreate");
    int rc = fork_aux(e, s);
    for (uint32_t i = 0; i < n_steps && rc == RAZ_OK; ++i) {
        for (int h = 0; h < H && rc == RAZ_OK; ++h)
            rc = launch_half_step(e, h, stream_of(e, h, s), &ev[3 * ((size_t)i * H + h)]);
    }
    const int rj = join_aux(e, s);
    hipError_t err = hipStreamSynchronize(s);
    if (rc == RAZ_OK && rj == RAZ_OK && err == hipSuccess) {
        for (size_t i = 0; i < (size_t)n_steps * H; ++i) {
            float a = 0.f, b = 0.f;
            hipEventElapsedTime(&a, ev[3 * i], ev[3 * i + 1]);
            hipEventElapsedTime(&b, ev[3 * i + 1], ev[3 * i + 2]);
            *tree_ms += a;
            *net_ms += b;
        }
    }
    for (auto& x : ev) hipEventDestroy(x);
    if (err != hipSuccess) return raz_fail_hip(err, "raz_engine_step_timed: sync");
    return rc != RAZ_OK ? rc : rj;
}

extern "C" int raz_engine_read_records(raz_engine* e, void* headers, uint32_t* root_n, double* root_w,
                                       uint32_t* n_plies, uint8_t* status, uint8_t* resigned,
                                       uint32_t* game_id, uint8_t* enable_resign, uint64_t* final_black,
                                       uint64_t* final_white, raz_stream_t stream) {
    if (!e) return raz_fail(RAZ_EINVAL, "raz_engine_read_records: NULL engine");
    const raz_engine_dev& d = e->dev;
    hipStream_t s = (hipStream_t)stream;
    const size_t B = d.B, MP = d.max_plies;
#define RAZ_D2H(dst, src, bytes)                                                              \
    if (dst) RAZ_HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s), "raz_engine_read_records")
    RAZ_D2H(headers, d.rec, B * MP * sizeof(raz_ply_header));
    RAZ_D2H(root_n, d.rec_n, B * MP * 64 * 4);
    if (root_w) {
        if (!d.rec_w) return raz_fail(RAZ_ESTATE, "raz_engine_read_records: engine created with record_root_w=0");
        RAZ_D2H(root_w, d.rec_w, B * MP * 64 * 8);
    }
    std::vector<raz_game> games(B);
    RAZ_D2H(games.data(), d.game, B * sizeof(raz_game));
#undef RAZ_D2H
    RAZ_HIP_TRY(hipStreamSynchronize(s), "raz_engine_read_records: sync");
    for (size_t g = 0; g < B; ++g) {
        const raz_game& G = games[g];
        if (n_plies) n_plies[g] = G.n_plies;
        if (status) status[g] = (uint8_t)G.status;
        if (resigned) {
            resigned[2 * g] = (uint8_t)G.resigned[0];
            resigned[2 * g + 1] = (uint8_t)G.resigned[1];
        }
        if (game_id) game_id[g] = G.game_id;
        if (enable_resign) enable_resign[g] = (uint8_t)G.enable_resign;
        if (final_black) final_black[g] = G.root_black;
        if (final_white) final_white[g] = G.root_white;
    }
    return RAZ_OK;
}

namespace {
// The largest n_plies over the slots [g0, g0 + n): one block.
__global__ __launch_bounds__(256) void k_records_extent(raz_engine_dev E, uint32_t g0, uint32_t n, uint32_t* out) {
    __shared__ uint32_t sh[256];
    uint32_t m = 0;
    for (uint32_t i = threadIdx.x; i < n; i += 256) {
        const uint32_t p = E.game[g0 + i].n_plies;
        m = p > m ? p : m;
    }
    sh[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] = sh[threadIdx.x] > sh[threadIdx.x + s] ? sh[threadIdx.x] : sh[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = sh[0];
}
// Records of slot g0 + blockIdx.x, cut to `plies` plies, into dense caller arrays (the unit of the record gather).
__global__ __launch_bounds__(256) void k_pack_records(raz_engine_dev E, uint32_t g0, uint32_t plies, raz_ply_header* hdr,
                                                      uint32_t* root_n, raz_game_summary* summ) {
    const uint32_t j = blockIdx.x, g = g0 + j;
    const raz_game& G = E.game[g];
    const uint32_t np = G.n_plies < plies ? G.n_plies : plies;
    const uint32_t* hs = (const uint32_t*)(E.rec + (size_t)g * E.max_plies);
    uint32_t* hd = (uint32_t*)(hdr + (size_t)j * plies);
    for (uint32_t i = threadIdx.x; i < plies * 12; i += 256) hd[i] = i < np * 12 ? hs[i] : 0u;
    const uint32_t* ns = E.rec_n + (size_t)g * E.max_plies * 64;
    uint32_t* nd = root_n + (size_t)j * plies * 64;
    for (uint32_t i = threadIdx.x; i < plies * 64; i += 256) nd[i] = i < np * 64 ? ns[i] : 0u;
    if (threadIdx.x == 0) {
        raz_game_summary S;
        S.final_black = G.root_black;
        S.final_white = G.root_white;
        S.game_id = G.game_id;
        S.n_plies = G.n_plies;
        S.status = (uint8_t)G.status;
        S.resigned_black = (uint8_t)G.resigned[0];
        S.resigned_white = (uint8_t)G.resigned[1];
        S.enable_resign = (uint8_t)G.enable_resign;
        S.sims_lo = (uint32_t)G.sims;
        summ[j] = S;
    }
}
}  // namespace

extern "C" int raz_engine_records_extent(raz_engine* e, uint32_t first_slot, uint32_t n_slots, uint32_t* max_plies,
                                         raz_stream_t stream) {
    if (!e || !max_plies) return raz_fail(RAZ_EINVAL, "raz_engine_records_extent: NULL argument");
    if (!e->started) return raz_fail(RAZ_ESTATE, "raz_engine_records_extent: call raz_engine_start first");
    if ((size_t)first_slot + n_slots > e->dev.B) return raz_fail(RAZ_EINVAL, "raz_engine_records_extent: slot range");
    hipStream_t s = (hipStream_t)stream;
    uint32_t* d = (uint32_t*)(e->dev.counters + 15);
    hipLaunchKernelGGL(k_records_extent, dim3(1), dim3(256), 0, s, e->dev, first_slot, n_slots, d);
    int rc = raz_check_launch("raz_engine_records_extent");
    if (rc != RAZ_OK) return rc;
    RAZ_HIP_TRY(hipMemcpyAsync(max_plies, d, 4, hipMemcpyDeviceToHost, s), "raz_engine_records_extent: copy");
    RAZ_HIP_TRY(hipStreamSynchronize(s), "raz_engine_records_extent: sync");
    return RAZ_OK;
}

extern "C" int raz_engine_pack_records(raz_engine* e, uint32_t first_slot, uint32_t n_slots, uint32_t plies,
                                       void* d_headers, uint32_t* d_root_n, raz_game_summary* d_summary,
                                       raz_stream_t stream) {
    if (!e || !d_headers || !d_root_n || !d_summary) return raz_fail(RAZ_EINVAL, "raz_engine_pack_records: NULL argument");
    if (!e->started) return raz_fail(RAZ_ESTATE, "raz_engine_pack_records: call raz_engine_start first");
    if ((size_t)first_slot + n_slots > e->dev.B || plies == 0 || plies > e->dev.max_plies)
        return raz_fail(RAZ_EINVAL, "raz_engine_pack_records: slot range / plies");
    if (n_slots == 0) return RAZ_OK;
    hipLaunchKernelGGL(k_pack_records, dim3(n_slots), dim3(256), 0, (hipStream_t)stream, e->dev, first_slot, plies,
                       (raz_ply_header*)d_headers, d_root_n, d_summary);
    return raz_check_launch("raz_engine_pack_records");
}

namespace {
// ---- continuous batching -------------------------------------------------------------------------------------------
// The reference worker starts its next game the moment one ends (worker/self_play.py:95-137).  Here a finished game's
// slot is emptied into the caller's outbox (row = game id - out_first_id: the outbox is in id order whatever order the
// games finish in) and restarted on the next unplayed id; ids are handed to the freed slots in slot order, results do not
// depend on the assignment (every random draw is keyed by the game id, the tree starts empty).
// plan[2g] = 0 nothing / 1 harvest, slot idles / 2 harvest and restart; plan[2g+1] = index of the new id.
__global__ __launch_bounds__(1024) void k_harvest_plan(raz_engine_dev E, uint32_t n_new, uint32_t out_first, uint32_t out_games,
                                                       uint32_t* plan, uint32_t* result) {
    __shared__ uint32_t sh[1024];
    __shared__ uint32_t base, skipped, live;
    if (threadIdx.x == 0) { base = 0; skipped = 0; live = 0; }
    __syncthreads();
    for (uint32_t g0 = 0; g0 < E.B; g0 += 1024) {
        const uint32_t g = g0 + threadIdx.x;
        bool fin = false;
        if (g < E.B) {
            const raz_game& G = E.game[g];
            fin = G.phase == RAZ_PHASE_DONE && G.status != 0;
            if (fin && G.game_id - out_first >= out_games) { fin = false; atomicAdd(&skipped, 1u); }   // no row for it: left in place
            if (!fin && G.phase != RAZ_PHASE_IDLE && G.phase != RAZ_PHASE_DONE) atomicAdd(&live, 1u);
        }
        sh[threadIdx.x] = fin ? 1u : 0u;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {   // inclusive scan
            const uint32_t v = (int)threadIdx.x >= off ? sh[threadIdx.x - off] : 0u;
            __syncthreads();
            sh[threadIdx.x] += v;
            __syncthreads();
        }
        const uint32_t rank = base + sh[threadIdx.x] - (fin ? 1u : 0u);
        if (g < E.B) {
            plan[2 * g] = fin ? (rank < n_new ? 2u : 1u) : 0u;
            plan[2 * g + 1] = rank;
        }
        __syncthreads();
        if (threadIdx.x == 1023) base += sh[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        result[0] = base;                          // harvested
        result[1] = base < n_new ? base : n_new;   // restarted
        result[2] = skipped;
        result[3] = live + result[1];              // slots playing after this call
    }
}

__global__ __launch_bounds__(256) void k_harvest_apply(raz_engine_dev E, const uint32_t* plan, uint32_t next_id, const uint32_t* sims,
                                                       const double* thr, uint32_t out_first, raz_ply_header* out_hdr,
                                                       uint32_t* out_n, raz_game_summary* out_sum, uint8_t* out_done) {
    const uint32_t g = blockIdx.x, what = plan[2 * g];
    if (what == 0) return;
    raz_game& G = E.game[g];
    const uint32_t row = G.game_id - out_first, np = G.n_plies < E.max_plies ? G.n_plies : E.max_plies, MP = E.max_plies;
    {   // the game's records -> its outbox row (unused plies zero)
        const uint32_t* hs = (const uint32_t*)(E.rec + (size_t)g * MP);
        uint32_t* hd = (uint32_t*)(out_hdr + (size_t)row * MP);
        for (uint32_t i = threadIdx.x; i < MP * 12; i += 256) hd[i] = i < np * 12 ? hs[i] : 0u;
        const uint32_t* ns = E.rec_n + (size_t)g * MP * 64;
        uint32_t* nd = out_n + (size_t)row * MP * 64;
        for (uint32_t i = threadIdx.x; i < MP * 64; i += 256) nd[i] = i < np * 64 ? ns[i] : 0u;
    }
    if (what == 2) {   // the slot's tree starts empty again
        raz_slot* tab = E.table + (size_t)g * E.H;
        for (uint32_t i = threadIdx.x; i < E.H; i += 256) tab[i].idx_tag = 0;
        if (E.M) {
            raz_slot* memo = E.memo + (size_t)g * E.M;
            for (uint32_t i = threadIdx.x; i < E.M; i += 256) memo[i].idx_tag = 0;
        }
        if (E.par) {
            uint32_t* sm = E.sim + (size_t)g * E.K * 64;
            for (uint32_t i = threadIdx.x; i < E.K * 64; i += 256) sm[i] = 0;
        }
        for (uint32_t i = threadIdx.x; i < E.K; i += 256) E.nn_active[(size_t)g * E.K + i] = 0;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    raz_game_summary S;
    S.final_black = G.root_black; S.final_white = G.root_white; S.game_id = G.game_id; S.n_plies = G.n_plies;
    S.status = (uint8_t)G.status; S.resigned_black = (uint8_t)G.resigned[0]; S.resigned_white = (uint8_t)G.resigned[1];
    S.enable_resign = (uint8_t)G.enable_resign; S.sims_lo = (uint32_t)G.sims;
    out_sum[row] = S;
    out_done[row] = 1;
    atomicAdd(&E.counters[8], 1ULL);
    atomicAdd(&E.counters[9], G.sims);
    atomicAdd(&E.counters[10], G.leaves);
    atomicAdd(&E.counters[11], G.selections);
    const uint32_t err = G.error;
    raz_game N;
    memset(&N, 0, sizeof N);
    N.error = err;   // (sticky: an overflowed pool must still be reported by raz_engine_stats_sync)
    N.root_black = RAZ_INIT_BLACK;
    N.root_white = RAZ_INIT_WHITE;
    N.player = RAZ_PLAYER_BLACK;
    N.leaf_kind = RAZ_LEAF_NONE;
    N.root_node = RAZ_NO_NODE;
    N.leaf_node = RAZ_NO_NODE;
    N.leaf_mirror = RAZ_NO_NODE;
    N.pool_used = 1;
    if (what == 2) {
        const uint32_t k = plan[2 * g + 1], id = next_id + k;
        N.phase = RAZ_PHASE_NEW_MOVE;
        N.game_id = id;
        double d0, d1;
        raz_rng_pair(E.cfg.seed, id, RAZ_RNG_GAME, 0, 0, 0, d0, d1);
        N.enable_resign = E.cfg.disable_resignation_rate <= d0 ? 1 : 0;  // worker/self_play.py:144
        N.sims_per_move = sims[k];
        if (thr) {
            const double t = thr[k];
            N.resign_mode = t == t ? 1u : 2u;   // NaN = resign_threshold is None
            N.resign_thr = (unsigned long long)__double_as_longlong(t);
        }
    } else {
        N.phase = RAZ_PHASE_IDLE;
        N.game_id = G.game_id;
    }
    G = N;
}
}  // namespace

extern "C" int raz_engine_harvest(raz_engine* e, uint32_t next_game_id, uint32_t n_new_ids, const uint32_t* sims_per_move,
                                  const double* resign_threshold, uint32_t out_first_id, uint32_t out_games, void* d_headers,
                                  uint32_t* d_root_n, raz_game_summary* d_summary, uint8_t* d_done, raz_harvest_result* result,
                                  raz_stream_t stream) {
    if (!e || !d_headers || !d_root_n || !d_summary || !d_done || !result || (n_new_ids && !sims_per_move))
        return raz_fail(RAZ_EINVAL, "raz_engine_harvest: NULL argument");
    if (!e->started) return raz_fail(RAZ_ESTATE, "raz_engine_harvest: call raz_engine_start first");
    const raz_engine_dev& d = e->dev;
    if (n_new_ids > d.B) n_new_ids = d.B;
    hipStream_t s = (hipStream_t)stream;
    if (n_new_ids) {
        RAZ_HIP_TRY(hipMemcpyAsync(e->d_sims, sims_per_move, (size_t)n_new_ids * 4, hipMemcpyHostToDevice, s), "raz_engine_harvest: copy sims");
        if (resign_threshold)
            RAZ_HIP_TRY(hipMemcpyAsync(e->d_thr, resign_threshold, (size_t)n_new_ids * 8, hipMemcpyHostToDevice, s), "raz_engine_harvest: copy thresholds");
    }
    uint32_t* plan = d.gc_remap;                       // scratch of k_gc, idle between steps: 2 words per slot
    uint32_t* dres = (uint32_t*)(d.counters + 12);     // 4 words
    hipLaunchKernelGGL(k_harvest_plan, dim3(1), dim3(1024), 0, s, d, n_new_ids, out_first_id, out_games, plan, dres);
    int rc = raz_check_launch("raz_engine_harvest: plan");
    if (rc != RAZ_OK) return rc;
    hipLaunchKernelGGL(k_harvest_apply, dim3(d.B), dim3(256), 0, s, d, (const uint32_t*)plan, next_game_id, (const uint32_t*)e->d_sims,
                       resign_threshold ? (const double*)e->d_thr : (const double*)nullptr, out_first_id, (raz_ply_header*)d_headers,
                       d_root_n, d_summary, d_done);
    rc = raz_check_launch("raz_engine_harvest: apply");
    if (rc != RAZ_OK) return rc;
    uint32_t h[4];
    RAZ_HIP_TRY(hipMemcpyAsync(h, dres, 16, hipMemcpyDeviceToHost, s), "raz_engine_harvest: copy result");
    RAZ_HIP_TRY(hipStreamSynchronize(s), "raz_engine_harvest: sync");   // (also keeps the host arrays alive long enough)
    result->harvested = h[0];
    result->restarted = h[1];
    result->skipped = h[2];
    result->playing = h[3];
    return RAZ_OK;
}

extern "C" size_t raz_leaf_cache_bytes(uint32_t log2_entries, size_t rows) {
    if (log2_entries < 10 || log2_entries > 28) return 0;
    return raz_leaf_cache_layout(log2_entries, rows, nullptr, nullptr);
}

extern "C" int raz_engine_set_leaf_cache(raz_engine* e, void* d_cache, size_t bytes, uint32_t log2_entries, uint32_t max_discs,
                                         raz_stream_t stream) {
    if (!e) return raz_fail(RAZ_EINVAL, "raz_engine_set_leaf_cache: NULL engine");
    drop_graph(e);   // a captured graph holds the old launch sequence
    if (!d_cache) {
        memset(&e->cache, 0, sizeof e->cache);
        return RAZ_OK;
    }
    const size_t rows = (size_t)e->dev.B * e->dev.K;
    const size_t need = raz_leaf_cache_bytes(log2_entries, rows);
    if (need == 0) return raz_fail(RAZ_EINVAL, "raz_engine_set_leaf_cache: log2_entries must be 10..28");
    if (bytes < need || ((uintptr_t)d_cache & 255)) return raz_fail(RAZ_ENOMEM, "raz_engine_set_leaf_cache: buffer too small (raz_leaf_cache_bytes) or not 256-byte aligned");
    raz_leaf_cache_dev c;
    raz_leaf_cache_layout(log2_entries, rows, (unsigned char*)d_cache, &c);
    c.max_discs = max_discs ? max_discs : 64u;
    const int rc = raz_leaf_cache_clear(c, rows, (hipStream_t)stream);
    if (rc != RAZ_OK) return rc;
    e->cache = c;
    return RAZ_OK;
}

extern "C" int raz_engine_leaf_cache_stats(raz_engine* e, uint64_t* out4, raz_stream_t stream) {
    if (!e || !out4) return raz_fail(RAZ_EINVAL, "raz_engine_leaf_cache_stats: NULL argument");
    if (!e->cache.tags) {
        memset(out4, 0, 32);
        return RAZ_OK;
    }
    RAZ_HIP_TRY(hipMemcpyAsync(out4, e->cache.counters, 32, hipMemcpyDeviceToHost, (hipStream_t)stream), "raz_engine_leaf_cache_stats: copy");
    RAZ_HIP_TRY(hipStreamSynchronize((hipStream_t)stream), "raz_engine_leaf_cache_stats: sync");
    return RAZ_OK;
}

extern "C" int raz_engine_set_resign_threshold(raz_engine* e, int has_threshold, double threshold) {
    if (!e) return raz_fail(RAZ_EINVAL, "raz_engine_set_resign_threshold: NULL engine");
    e->dev.cfg.has_resign_threshold = has_threshold ? 1 : 0;
    e->dev.cfg.resign_threshold = threshold;
    drop_graph(e);   // a captured graph holds the old parameter block
    return RAZ_OK;
}

extern "C" void* raz_engine_device_ptr(raz_engine* e, int which) {
    if (!e) return nullptr;
    switch (which) {
        case 0: return e->dev.rec;
        case 1: return e->dev.rec_n;
        case 2: return e->dev.rec_w;
        case 3: return e->dev.game;
        case 5: return e->dev.prof;
        default: return nullptr;
    }
}
