// raz_engine_core.h — the per-game device code of the engine (one wavefront per game, lane = board square): wave helpers,
// tree storage, the control block in registers, PUCT, the end-game solver, backup, the per-move controller and the descent.
// Included by raz_engine.hip (k_tree, k_tree_par and the rest of the engine) and by raz_engine_fused.hip (k_tree_net): two
// translation units, so that adding the fused kernel leaves the code generated for the others untouched.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
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
// the same where one lane reads what ANOTHER lane of the wave has just written and no cross-lane instruction lies in between: the wave
// barrier costs nothing on the hardware (one wave executes in lock-step) and is the meeting point of the lanes on the wave emulator
__device__ __forceinline__ void wave_sync_lanes() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ------------------------------------------------------------------ words that cross between CONCURRENT kernels
// The solver pool's round may run on its own stream beside the tree launches (raz_engine.hip, pool_every > 1), so tree kernels and pool
// kernels exchange a few words with no kernel boundary in between: a game's request header (raz_solve_hdr: position + state), the
// answer word, memo entries.  On MI355X the per-XCD L2s are not coherent with each other and a CU's L1 is never refreshed by another
// CU's stores (MI355X_MICROARCH.md "inter-workgroup visibility"): such words are written AND read with agent-scope atomics only
// (sc1: stores write through, loads bypass L1) - the guide's "8-byte agent atomics on both sides" form - and a word that publishes
// others (a memo slot's tag, a request's state) is stored after the publisher's `s_waitcnt vmcnt(0)` (xk_drain: the asm form, which the
// compiler cannot drop).  tools/litmus_slot_handoff.hip is the hardware check of exactly this against the plain-store / plain-load form
// of rounds 1-5 (profiles/r6/litmus_slot_handoff.json).  No cache-wide fence (buffer_wbl2 / buffer_inv) is needed or issued.
// The TREE kernels' side of the exchange always uses the agent-scope forms (a look at a request header per launch, the memo probe
// before a request, the request's four stores: nothing that is timed).  The POOL kernels choose at run time:
// `conc` (raz_engine_dev.xk) is 0 when the pool's round runs between the tree launches on their own stream (pool_every == 1, whole-game
// batches): every hand-off then crosses a kernel boundary and plain accesses are what they always were - lines stay in L1 / L2, which
// is worth 4-5 % on mini.yml as shipped (profiles/r6/handoff_*_ab*.jsonl: agent-scope stores drop the line from the XCD's L2, and the
// one look a suspended game takes at its request header then comes from memory).
// (The "plain" forms are wavefront-scope relaxed atomics: the same instructions as an ordinary load / store - no sc bits, served by
// L1 / L2 - but always VECTOR memory instructions: with a real plain load of a uniform address in one arm the compiler builds a
// scalar load there, a vector load in the other, and fails on the merge in the fused kernels - "illegal VGPR to SGPR copy".)
__device__ __forceinline__ void xk_store64(bool conc, unsigned long long* p, unsigned long long v) {
    if (conc) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}
__device__ __forceinline__ void xk_store32(bool conc, uint32_t* p, uint32_t v) {
    if (conc) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}
__device__ __forceinline__ unsigned long long xk_load64(bool conc, const unsigned long long* p) {
    unsigned long long v = conc ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
#ifndef RAZ_WAVE_EMU
    asm volatile("" : "+v"(v));   // (the merged value stays a vector register: see above)
#endif
    return v;
}
__device__ __forceinline__ uint32_t xk_load32(bool conc, const uint32_t* p) {
    uint32_t v = conc ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
#ifndef RAZ_WAVE_EMU
    asm volatile("" : "+v"(v));
#endif
    return v;
}
__device__ __forceinline__ void xk_drain(bool conc) {
#ifndef RAZ_WAVE_EMU
    if (conc) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}

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
    const uint32_t h = key_hash(b, w, tagkey) & ~(uint32_t)(RAZ_TABLE_PROBE - 1);   // aligned groups: one 128-byte line per request
    Found f;
    f.found = false;
    f.node = 0;
    f.slot = 0xffffffffu;
    const bool probing = lane < RAZ_TABLE_PROBE;   // (the other lanes request nothing)
    for (uint32_t r = 0; r < E.H; r += RAZ_TABLE_PROBE) {
        const uint32_t si = (h + r + (uint32_t)(lane & (RAZ_TABLE_PROBE - 1))) & mask;
        const raz_slot* s = tab + si;
        raz_bb sb = 0, sw = 0;
        uint32_t it = RAZ_SLOT_USED, lk = 0;
        if (probing) {
            sb = s->black;
            sw = s->white;
            it = s->idx_tag;
            lk = s->link;
        }
        const bool used = (it & RAZ_SLOT_USED) != 0;
        const bool match = probing && used && sb == b && sw == w && (it & RAZ_SLOT_KEYMASK) == tagkey;
        const unsigned long long mm = __ballot(match) & ((1ULL << RAZ_TABLE_PROBE) - 1ULL);
        const unsigned long long em = __ballot(probing && !used) & ((1ULL << RAZ_TABLE_PROBE) - 1ULL);
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

// The engine's descriptor (raz_engine_dev: ~90 scalar registers of pointers and configuration words) as a kernel ARGUMENT is loaded once
// at the kernel's entry and then lives in scalar registers for the whole launch - more than a wave has, so the compiler parks them in the
// lanes of spare vector registers (v_writelane / v_readlane: vector-ALU work in an issue-bound kernel).  Handing the step loop the
// kernel-argument segment's address through an empty asm once per step makes every use a scalar load from the (cached) segment
// where it is needed instead.  k_tree_net / k_tree_par_net: bit 3 of RAZ_FRESH_1 / RAZ_FRESH_K; k_tree / k_tree_par: RAZ_FRESH_DESC.
#ifndef RAZ_FRESH_DESC
#define RAZ_FRESH_DESC 1
#endif
template <int ON>
__device__ __forceinline__ const raz_engine_dev& fresh_descriptor(const raz_engine_dev& E) {
#ifndef RAZ_WAVE_EMU
    if (ON) {
        typedef const raz_engine_dev __attribute__((address_space(4))) * kernarg_ptr;
        kernarg_ptr p = (kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();   // (the descriptor is the kernels' FIRST argument)
        asm volatile("" : "+s"(p));
        return *(const raz_engine_dev*)p;
    }
#endif
    return E;
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
    uint32_t solve_pending;         // an end-game solve waits for the solver pool: begin_move's (the phase stays NEW_MOVE) or one
                                    // inside a simulation (leaf_kind RAZ_LEAF_SOLVE_PENDING); the game does nothing more in this launch
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
// disc difference, strict improvement keeps the first maximum.  Three places compute it:
//   * positions with <= RAZ_SOLVER_SCALAR_EMPTIES empties: right here, one wave-uniform DFS on the scalar unit, frames in LDS;
//   * larger ones (<= RAZ_SOLVER_MAX_DEPTH empties): the tree kernel only POSTS the position in its game's request block and
//     suspends (RAZ_SOLVE_PENDING); the solver pool (raz_solver_pool.h: k_solve_scan / k_solve_run, launched between tree
//     launches) answers it, and the next call for the same position returns the answer;
//   * a per-game memo in HBM (positions with >= 4 empties; it only saves time, exactly as the reference's dict does) is shared by
//     all of them.
// f is a function of the position, so WHEN an answer arrives changes nothing but a game's wall time.
struct SolverLDS {   // the scalar search's frames (depth <= RAZ_SOLVER_SCALAR_EMPTIES + passes; positions of > RAZ_SOLVER_MAX_DEPTH empties never get here: validate())
    unsigned long long own[16], enemy[16], left[16];
    int best_move[16], best_score[16], paction[16], flip[16], fresh[16];
};
#define RAZ_SOLVER_LDS_BYTES 704
static_assert(sizeof(SolverLDS) == RAZ_SOLVER_LDS_BYTES, "SolverLDS layout");

// A memo slot is written ONCE between two clears of the table: a writer CLAIMS it (idx_tag 0 -> RAZ_MEMO_CLAIMED, one atomic, so
// of several lanes - of this wave or of any worker wave of the pool - exactly one wins), stores the key, then publishes the tag
// (bit 31).  Readers take a slot whose bit 31 is clear for empty, so a claimed slot is a miss, never a torn entry.
#define RAZ_MEMO_CLAIMED 0x20000000u
// Writers and readers may be waves of concurrent kernels on different XCDs (the pool's round beside the tree launches): key words and
// tag are agent-scope atomics on both sides, the tag stored after the key stores have drained (xk_* above).  A reader that still gets
// the tag before the keys compares unequal: a miss.
__device__ __forceinline__ bool memo_claim_and_write(bool conc, raz_slot* s, raz_bb own, raz_bb enemy, uint32_t tag) {
    if (atomicCAS(&s->idx_tag, 0u, RAZ_MEMO_CLAIMED) != 0u) return false;
    xk_store64(conc, &s->black, own);
    xk_store64(conc, &s->white, enemy);
    if (conc) xk_drain(true);
    else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // (compiler order only)
    __hip_atomic_store(&s->idx_tag, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
}

__device__ bool memo_find(const raz_engine_dev& E, uint32_t g, raz_bb own, raz_bb enemy, uint32_t exact, int lane,
                          int& move, int& score) {
    const raz_slot* tab = E.memo + (size_t)g * E.M;
    const uint32_t mask = E.M - 1;
    const uint32_t h = key_hash(own, enemy, 8u + exact);
    for (uint32_t r = 0; r < 64; r += RAZ_PROBE) {
        const raz_slot* s = tab + ((h + r + (uint32_t)(lane & (RAZ_PROBE - 1))) & mask);
        constexpr bool conc = true;   // (tree side: always the agent-scope forms, see xk_* above)
        const uint32_t it = xk_load32(conc, &s->idx_tag);
        const raz_bb sb = xk_load64(conc, &s->black), sw = xk_load64(conc, &s->white);
        const bool used = (it >> 31) != 0;
        const bool match = used && sb == own && sw == enemy && ((it >> 30) & 1u) == exact;
        const unsigned long long mm = __ballot(match) & 0xffffULL;
        const unsigned long long em = __ballot(it == 0u) & 0xffffULL;
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
    const uint32_t tag = 0x80000000u | (exact << 30) | ((uint32_t)(move + 1) << 8) | (uint32_t)(score + 128);
    for (uint32_t r = 0; r < 64; r += RAZ_PROBE) {
        const uint32_t si = (h + r + (uint32_t)(lane & (RAZ_PROBE - 1))) & mask;
        unsigned long long em = __ballot(xk_load32(true, &tab[si].idx_tag) == 0u) & 0xffffULL;
        while (em) {   // (wave-uniform) the first free slot of the group - or the next one, if a worker wave of the pool claimed it meanwhile
            const int j = __ffsll((long long)em) - 1;
            em &= em - 1;
            uint32_t ok = 0u;
            if (lane == 0) ok = memo_claim_and_write(true, tab + ((h + r + (uint32_t)j) & mask), own, enemy, tag) ? 1u : 0u;
            if (uni(ok)) {
                wave_sync();
                return;
            }
        }
    }  // 64 occupied slots in a row: the memo is (locally) full; skipping the insert only costs time
}

// the memo, one lane on its own (the pool's workers, the scans of k_solve_scan): the key's home slot and the next one
__device__ bool memo_find_lane(const raz_engine_dev& E, uint32_t g, raz_bb own, raz_bb enemy, uint32_t exact, int& move, int& score) {
    const raz_slot* tab = E.memo + (size_t)g * E.M;
    const uint32_t mask = E.M - 1, h = key_hash(own, enemy, 8u + exact);
    // both slots' fields are requested together: ONE round trip into the game's table (the 64 searches of a worker wave advance in
    // lockstep, so a dependent second request would be paid by all of them)
    const raz_slot *s0 = tab + (h & mask), *s1 = tab + ((h + 1u) & mask);
    const bool conc = E.xk != 0u;
    const uint32_t it0 = xk_load32(conc, &s0->idx_tag), it1 = xk_load32(conc, &s1->idx_tag);
    const raz_bb b0 = xk_load64(conc, &s0->black), w0 = xk_load64(conc, &s0->white), b1 = xk_load64(conc, &s1->black), w1 = xk_load64(conc, &s1->white);
    uint32_t it;
    if (!(it0 >> 31)) return false;
    if (b0 == own && w0 == enemy && ((it0 >> 30) & 1u) == exact)
        it = it0;
    else if ((it1 >> 31) && b1 == own && w1 == enemy && ((it1 >> 30) & 1u) == exact)
        it = it1;
    else
        return false;
    move = (int)((it >> 8) & 0xffu) - 1;
    score = (int)(it & 0xffu) - 128;
    return true;
}
__device__ void memo_put_lane(const raz_engine_dev& E, uint32_t g, raz_bb own, raz_bb enemy, uint32_t exact, int move, int score) {
    raz_slot* tab = E.memo + (size_t)g * E.M;
    const uint32_t mask = E.M - 1, h = key_hash(own, enemy, 8u + exact);
    const uint32_t tag = 0x80000000u | (exact << 30) | ((uint32_t)(move + 1) << 8) | (uint32_t)(score + 128);
    for (uint32_t r = 0; r < 2; ++r)
        if (memo_claim_and_write(E.xk != 0u, tab + ((h + r) & mask), own, enemy, tag)) return;
    // both slots taken: skipping the insert only costs time
}

// ReversiSolver.solve for the side to move (own, enemy), wave-uniform: ONE depth-first search on the scalar unit.  Returns false
// for the reference's (None, None) (no legal move at the root: never the case for a running game).  Used for the smallest trees
// (<= RAZ_SOLVER_SCALAR_EMPTIES empties); the pool takes the larger ones.
__device__ bool solver_solve_scalar(const raz_engine_dev& E, uint32_t g, int lane, raz_bb own0, raz_bb enemy0, uint32_t exact,
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

#define RAZ_SOLVER_SCALAR_EMPTIES 4   // (round 4: 6 - a wave-uniform search of 720 leaf paths with a memo probe per node was the tree kernel's straggler, up to 3 ms)
#define RAZ_SOLVER_MAX_DEPTH 14
#define RAZ_SOLVE_NONE 0
#define RAZ_SOLVE_DONE 1
#define RAZ_SOLVE_PENDING 2
// the tree kernels: 4 waves per SIMD whether the solver is compiled in or not (the lane-parallel search that needed 256 registers in
// round 4 now lives in the pool's own kernel)
#ifdef RAZ_WAVE_EMU
#define RAZ_TREE_WAVES(SOLVER)
#else
#define RAZ_TREE_WAVES(SOLVER) __attribute__((amdgpu_waves_per_eu(4, 4)))
#endif

__device__ __forceinline__ raz_solve_hdr* solve_hdr(const raz_engine_dev& E, uint32_t g) {
    return (raz_solve_hdr*)(E.solver_ws + (size_t)g * RAZ_SOLVER_WS_BYTES);
}
// A game whose request is still with the pool has nothing to do in this launch: one word tells (the tree kernels look at it before
// they load anything else - in a solver-bound batch most games of a launch are in that state).
__device__ __forceinline__ bool solve_in_flight(const raz_engine_dev& E, uint32_t g) {
    const uint32_t st = RAZ_SOLVE_STATE(uni(xk_load32(true, &solve_hdr(E, g)->state)));   // (the pool may publish the answer while this kernel runs)
    return st == RAZ_SOLVE_REQUESTED || st == RAZ_SOLVE_RUNNING;
}

// f_mode(own0, enemy0) for the game's wave: RAZ_SOLVE_DONE (out_move / out_score), RAZ_SOLVE_NONE (the reference's (None, None)) or
// RAZ_SOLVE_PENDING - the position now stands in the game's request block and the caller suspends: begin_move leaves the phase at
// NEW_MOVE, a descent stops where it stands (select_leaf: RAZ_LEAF_SOLVE_PENDING); both call again at the next launch.  A game
// has ONE request in flight: it does nothing else until the answer is there.
__device__ int solver_solve(const raz_engine_dev& E, uint32_t g, int lane, raz_bb own0, raz_bb enemy0, uint32_t exact,
                            SolverLDS* S, int& out_move, int& out_score) {
    const int empties = bb_popcount(~(own0 | enemy0));
    if (empties <= RAZ_SOLVER_SCALAR_EMPTIES || empties > RAZ_SOLVER_MAX_DEPTH)
        return solver_solve_scalar(E, g, lane, own0, enemy0, exact, S, out_move, out_score) ? RAZ_SOLVE_DONE : RAZ_SOLVE_NONE;
    {
        int rm, rs;
        wave_sync();
        if (memo_find(E, g, own0, enemy0, exact, lane, rm, rs)) {
            out_move = rm;
            out_score = rs;
            return rm >= 0 ? RAZ_SOLVE_DONE : RAZ_SOLVE_NONE;
        }
    }
    raz_solve_hdr* h = solve_hdr(E, g);
    // every word of the header that the other side reads or writes while this kernel runs goes through xk_* (state and gen share one
    // aligned 8-byte word: the pool's kernels see a request's generation and state together)
    constexpr bool conc = true;   // (tree side: always the agent-scope forms)
    const unsigned long long sg = uni((raz_bb)xk_load64(conc, (const unsigned long long*)h));
    const uint32_t word = (uint32_t)sg, st = RAZ_SOLVE_STATE(word), gen = (uint32_t)(sg >> 32);
    const bool same = uni((uint32_t)(xk_load64(conc, &h->own0) == own0 && xk_load64(conc, &h->enemy0) == enemy0 && xk_load32(conc, &h->exact) == exact)) != 0u;
    if (same && st == RAZ_SOLVE_ANSWERED) {   // (the block keeps the last answer: the memo may have had no room for it)
        out_move = RAZ_SOLVE_ANSWER_MOVE(word);
        out_score = RAZ_SOLVE_ANSWER_SCORE(word);
        return (int)RAZ_SOLVE_ANSWER_KIND(word);
    }
    if (same && (st == RAZ_SOLVE_REQUESTED || st == RAZ_SOLVE_RUNNING)) return RAZ_SOLVE_PENDING;
    wave_sync();
    if (lane == 0) {   // a new request (an abandoned one - another position - is overwritten: its workers see the new gen and drop their tasks)
        // the position first, drained; then {state, gen} in ONE 8-byte store: a scan that sees REQUESTED sees this request's position
        // (five plain stores could be merged or reordered by the compiler, and sat in this XCD's L2 until the kernel ended)
        xk_store64(conc, &h->own0, own0);
        xk_store64(conc, &h->enemy0, enemy0);
        xk_store32(conc, &h->exact, exact);
        xk_drain(conc);
        xk_store64(conc, (unsigned long long*)h, (unsigned long long)RAZ_SOLVE_REQUESTED | ((unsigned long long)(gen + 1u) << 32));
        h->posted += 1u;   // (this side's word: nobody else touches it while a kernel runs)
    }
    wave_sync();
    return RAZ_SOLVE_PENDING;
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
            wave_sync_lanes();   // lane 0 adds to cells that OTHER lanes have just initialised (place_leaf: a brand-new node or mirror)
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
        const int solved = solver_solve(E, g, lane, own, enemy, 1u, S, sm, ss);
        if (solved == RAZ_SOLVE_PENDING) {   // the search is parked in the game's workspace; this launch is over for the game
            R.solve_pending = 1u;
            return;
        }
        if (solved == RAZ_SOLVE_DONE) {
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
// A solve inside the simulation (:237-251) that runs out of the launch's solver budget SUSPENDS the descent: leaf_kind
// RAZ_LEAF_SOLVE_PENDING, leaf_node / depth = where it stands, the path so far stored like any other.  The caller ends the game's
// launch there and calls again at the next one with start_node / start_depth = that place: the solve goes on from its parked state
// (it is a function of the position alone) and the descent from its answer, so nothing but wall time depends on the budget.  A
// solve on a FIRST arrival has no node to stand on: it stands on the parent with forced_rank = the edge it took (selected, given
// its virtual loss and put on the path exactly once, before the suspension), replays the move and meets the same solve again.
template <bool SOLVER, bool PAR>
__device__ void select_leaf(const raz_engine_dev& E, Regs& R, uint32_t g, int lane, SolverLDS* S, uint32_t nn_index,
                            uint32_t start_node, int start_depth, bool polling, int forced_rank) {
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
    int depth = start_depth;
    uint32_t kind = RAZ_LEAF_NONE;
    uint32_t node = start_node;  // always exists (begin_move)
    const int first_depth = start_depth;
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
        if (SOLVER && t_insim && !(PAR && polling) && forced_rank < 0 && bb_popcount(env.black | env.white) - 4 >= t_insim) {  // solver inside simulations (:237-251)
            const raz_bb so = env.np == 1 ? env.black : env.white, se = env.np == 1 ? env.white : env.black;
            int sm, ss;
            const int solved = solver_solve(E, g, lane, so, se, 0u, S, sm, ss);
            if (solved == RAZ_SOLVE_PENDING) {
                kind = RAZ_LEAF_SOLVE_PENDING;
                leaf_node = node;
                break;
            }
            if (solved == RAZ_SOLVE_DONE && sm != 0) {  // `if action:` ignores square 0
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
        if (PAR && forced_rank < 0) {
            polling = false;
            if ((uni(tag) >> (6 + pl)) & 1u) {  // while key in self.now_expanding: await asyncio.sleep(...) (:253-254)
                kind = RAZ_LEAF_PARKED;
                leaf_node = node;
                break;
            }
        }
        if (forced_rank < 0 && !((uni(tag) >> (4 + pl)) & 1u)) {  // key not in this player's `expanded` (:257)
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
        const bool replay = SOLVER && forced_rank >= 0;   // (the edge was chosen before the suspension: no second draw, no second virtual loss)
        const int r = replay ? forced_rank : select_action(E, R, g, Wi, Ni, Pi, L, env.np, depth == 0, game_id, lane);   // rank of the move
        const int a = square_of_rank(env.legal, r, lane);
        forced_rank = -1;
        if (PAR && lane == r && !replay) {  // var_n[key][action_t] += virtual_loss; var_w[key][action_t] -= virtual_loss_for_w (:270-271)
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
            const int solved = solver_solve(E, g, lane, so, se, 0u, S, sm, ss);
            if (solved == RAZ_SOLVE_PENDING) {   // stand on the parent, the edge taken
                kind = RAZ_LEAF_SOLVE_PENDING;
                leaf_node = node;
                solved_action = r + 1;
                --depth;
                break;
            }
            if (solved == RAZ_SOLVE_DONE && sm != 0) {
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
    if (SOLVER && kind == RAZ_LEAF_SOLVE_PENDING) {
        S32(R, GW(leaf_action), (uint32_t)solved_action);
        S32(R, GW(leaf_node), leaf_node);
        R.solve_pending = 1u;
    }
    if (PAR && kind == RAZ_LEAF_PARKED) S32(R, GW(sim_parked), leaf_node);
    S32(R, GW(leaf_term_v), __float_as_uint(term_v));
    S32(R, GW(leaf_kind), kind);
    S32(R, GW(depth), (uint32_t)depth);
    ADD64(R, GW(selections), (raz_bb)(depth - first_depth));
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

// ------------------------------------------------------------------ simulation slots (parallel_search_num > 1: k_tree_par, k_tree_par_net)
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

}  // namespace
