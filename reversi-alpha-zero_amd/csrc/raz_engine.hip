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

#include "raz_engine_core.h"   // the per-game device code: helpers, PUCT, solver requests, backup, controller, descent
#include "raz_solver_pool.h"   // the end-game solver's worker pool: k_solve_scan, k_solve_run

namespace {

// ------------------------------------------------------------------ the tree kernel
// SOLVER = false compiles the end-game solver's call sites (and the scalar search's LDS frames) out: the common case, and the
// bench configuration.  With the solver in, the kernel only solves the smallest positions itself (<= RAZ_SOLVER_SCALAR_EMPTIES = 4 empties, wave-uniform);
// larger ones are posted to the solver pool (raz_solver_pool.h) and the game suspends until the answer is there.
template <bool SOLVER>
__global__ __launch_bounds__(64) RAZ_TREE_WAVES(SOLVER) void k_tree(raz_engine_dev E, uint32_t g0, uint32_t count) {
    if (blockIdx.x >= count) return;
    __shared__ float lds64[64];
    __shared__ SolverLDS slds_store;
    SolverLDS* slds_p = SOLVER ? &slds_store : nullptr;
    const uint32_t g = g0 + blockIdx.x;
    const int lane = threadIdx.x;
    if (g >= E.B) return;
    if (SOLVER && solve_in_flight(E, g)) {
        if (lane == 0) E.nn_active[g] = 0;
        return;
    }
    // ONE round trip: control block, path, the net's answer for the leaf of the previous launch
    uint32_t* gw = (uint32_t*)(E.game + g);
    Regs R;
    R.cw = gw[lane];
    path_load(E, R, (size_t)g, lane, true);
    R.pol_raw = E.nn_policy[(size_t)g * 64 + lane];
    R.val = E.nn_value[g];
    R.nn = 0u;
    R.path_dirty = 0u;
    R.solve_pending = 0u;
    path_load_rest(E, R, (size_t)g, lane);
    {
        const uint32_t phase = G32(R, GW(phase));
        if (phase == RAZ_PHASE_DONE || phase == RAZ_PHASE_IDLE || G32(R, GW(error))) return;  // nn_active is already 0
    }
    if (RAZ_PROF_ON(E) && lane == 0) E.prof[(size_t)g * 8 + 5] += 1;
    const int inner_max = ((E.cfg.reserved >> 12) & 0xf) ? (int)((E.cfg.reserved >> 12) & 0xf) : kInnerMax;
    const raz_engine_dev& E0 = E;
    for (int it = 0; it < inner_max; ++it) {
        const raz_engine_dev& E = fresh_descriptor<RAZ_FRESH_DESC>(E0);
        uint32_t phase = G32(R, GW(phase));
        if (phase == RAZ_PHASE_DONE || phase == RAZ_PHASE_IDLE) break;
        if (G32(R, GW(error))) break;
        // a descent suspended at an in-simulation solve (select_leaf) goes on where it stands, before anything else
        const bool suspended = SOLVER && G32(R, GW(leaf_kind)) == RAZ_LEAF_SOLVE_PENDING;
        unsigned long long t0 = prof_now();
        if (!suspended) {
            if (G32(R, GW(leaf_kind)) != RAZ_LEAF_NONE) backup_leaf<false>(E, R, g, G32(R, GW(player)) - 1, lane, lds64);
            prof_add(E, g, 0, t0, lane);
            t0 = prof_now();
            // controller: loop because a decided move may immediately need another decision
            // (turn-0 bypass) before a search with simulations starts
            for (int guard = 0; guard < 8; ++guard) {
                phase = G32(R, GW(phase));
                if (phase == RAZ_PHASE_NEW_MOVE) {
                    if (R.solve_pending) break;   // the root's end-game solve ran out of this launch's budget: it goes on at the next launch
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
        }
        select_leaf<SOLVER, false>(E, R, g, lane, slds_p, g, suspended ? G32(R, GW(leaf_node)) : G32(R, GW(root_node)),
                                   suspended ? (int)G32(R, GW(depth)) : 0, false, suspended ? (int)G32(R, GW(leaf_action)) - 1 : -1);
        prof_add(E, g, 2, t0, lane);
        const uint32_t lk = G32(R, GW(leaf_kind));
        if (lk != RAZ_LEAF_TERMINAL && lk != RAZ_LEAF_SOLVED) break;  // needs the net (or, suspended at a solve, the next launch)
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
// The kernel is ONE loop with a single call site each for the controller, the descent and the return
// path (the three large inlined bodies): duplicating them per phase doubled the code to 66 KB, more
// than the instruction cache two CUs share.  Each iteration picks the next operation of the round -
// resume the oldest queued leaf (B), start a simulation in a free slot (C, C'), wake the next sleeper
// (D) - then runs at most one slot load, one descent, one return.
template <bool SOLVER>
__global__ __launch_bounds__(64) RAZ_TREE_WAVES(SOLVER) void k_tree_par(raz_engine_dev E, uint32_t g0, uint32_t count) {
    if (blockIdx.x >= count) return;
    __shared__ float lds64[64];
    __shared__ SolverLDS slds_store;
    SolverLDS* slds_p = SOLVER ? &slds_store : nullptr;
    const uint32_t g = g0 + blockIdx.x;
    const int lane = threadIdx.x;
    if (g >= E.B) return;
    const uint32_t K = E.K;
    if (SOLVER && solve_in_flight(E, g)) {   // (leaves queued before the solve was posted have been evaluated: their flags go down)
        if (lane < (int)K) E.nn_active[(size_t)g * K + lane] = 0;
        return;
    }
    const unsigned long long kmask = (1ULL << K) - 1ULL;  // K <= 16
    uint32_t* gw = (uint32_t*)(E.game + g);
    Regs R;
    R.cw = gw[lane];
    R.pnode = R.pmirror = R.pact = 0u;
    R.pol_raw = 0.0f;
    R.val = 0.0f;
    R.nn = 0u;
    R.path_dirty = 0u;
    R.solve_pending = 0u;
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
    constexpr uint32_t kStageB = 0u, kStageC = 1u, kStageC2 = 2u, kStageD = 4u;  // D outlives a launch only under a suspended solve
    uint32_t stage = G32(R, GW(par_stage));
    unsigned long long dmask = stage == kStageD ? (unsigned long long)G32(R, GW(par_dmask)) : 0ULL;  // sleepers still to poll in D
    const raz_engine_dev& E0 = E;
    for (;;) {
        const raz_engine_dev& E = fresh_descriptor<RAZ_FRESH_DESC>(E0);
        if (G32(R, GW(error))) break;
        // ---- the next operation of the round
        int j = -1;
        bool resume = false, wake = false, suspended = false;
        const unsigned long long solving = SOLVER ? (__ballot(T.st == RAZ_SIM_SOLVING) & kmask) : 0ULL;
        if (solving) {  // a descent suspended at an in-simulation solve goes on first: it was a start (C / C') or a wake (D)
            j = __ffsll((long long)solving) - 1;
            suspended = true;
            wake = stage == kStageD;
        } else if (stage == kStageB) {
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
                    if (R.solve_pending) break;   // the root's end-game solve ran out of this launch's budget: it goes on at the next launch
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
        if (resume || wake || suspended) slot_load(E, R, g, (uint32_t)j, lane, resume);
        if (!resume) {
            const unsigned long long t0 = prof_now();
            select_leaf<SOLVER, true>(E, R, g, lane, slds_p, g * K + (uint32_t)j,
                                      suspended ? G32(R, GW(leaf_node)) : (wake ? lane_u32(T.pk, j) : G32(R, GW(root_node))),
                                      (wake || suspended) ? (int)G32(R, GW(depth)) : 0, wake && !suspended,
                                      suspended ? (int)G32(R, GW(leaf_action)) - 1 : -1);
            prof_add(E, g, 2, t0, lane);
            const uint32_t kind = G32(R, GW(leaf_kind));
            if (SOLVER && kind == RAZ_LEAF_SOLVE_PENDING) {  // out of solver budget: the slot keeps the descent, the launch is over for the game
                T.st = writelane_r(T.st, RAZ_SIM_SOLVING, j, lane);
                slot_store(E, R, g, (uint32_t)j, lane);
                S32(R, GW(leaf_kind), RAZ_LEAF_NONE);
                break;
            }
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
    if (SOLVER && R.solve_pending && stage == kStageD) {
        S32(R, GW(par_dmask), (uint32_t)dmask);
        S32(R, GW(par_stage), kStageD);
    } else
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
            if (lk == RAZ_LEAF_SOLVE_PENDING && ln != RAZ_NO_NODE) blk[GW(leaf_node)] = gc_translate(E, g, remap, ln);   // the node a suspended descent stands on
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
        uint32_t si = key_hash(kb, kw, tagkey) & mask & ~(uint32_t)(RAZ_TABLE_PROBE - 1);   // table_find's order: the aligned group first
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

// Worker waves of the solver pool: the caller's figure, or one per four games, at most 1280 - five waves per CU is what a CU's LDS holds
// of their 28 KB of frames, so 1280 are resident at once.  Measured on mini.yml as shipped (4096 games, profiles/r5/solver_pool_*):
// with the fourth ply of tasks 1024 waves are the best size, 512 and 1280 within 4 %; beyond 1280 the launch runs in two rounds.
// At least kPoolParts, so that every slice of the batch has a wave.
constexpr uint32_t kPoolParts = 8;   // == kMaxParts below
inline uint32_t pool_waves_of(const raz_engine_config& cfg) {
    uint32_t w = cfg.solver_pool_waves ? cfg.solver_pool_waves : (cfg.n_games + 3) / 4;
    if (w > 1280u && !cfg.solver_pool_waves) w = 1280u;
    if (w > 16384u) w = 16384u;
    return w < kPoolParts ? kPoolParts : w;
}

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
    d.solver_ws = cfg.solver_memo_slots ? take(B * (size_t)RAZ_SOLVER_WS_BYTES) : nullptr;
    d.W = cfg.solver_memo_slots ? pool_waves_of(cfg) : 0u;
    d.pool_hdr = (raz_solver_pool_hdr*)take(d.W ? kPoolParts * sizeof(raz_solver_pool_hdr) : 0);
    d.pool_active = (uint32_t*)take(d.W ? B * 4 : 0);
    d.pool_state = (unsigned long long*)take((size_t)d.W * RAZ_SOLVER_WORKER_STATE_BYTES);
    d.pool_frames = (unsigned long long*)take((size_t)d.W * RAZ_SOLVER_WORKER_FRAME_BYTES);
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
    if ((cfg->reserved & ~0x0fffff1bu) || cfg->reserved2)
        return raz_fail(RAZ_EINVAL, "raz_engine: reserved bits 2, 5-7 and 28-31 and reserved2 must be 0 (ABI 3: bit 2 - hipGraph replay - is gone)");
    if (cfg->share_mtcs_info && !cfg->mirror_updates)
        return raz_fail(RAZ_EINVAL, "raz_engine: share_mtcs_info=1 requires mirror_updates=1 (player.py:279-280)");
    return RAZ_OK;
}

}  // namespace

constexpr int kMaxParts = 8;
static_assert(kMaxParts == (int)kPoolParts, "one pool header per slice");

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
static hipError_t raz_pool_stream(int h, hipStream_t* out) {   // h < kMaxParts: a slice's stream; kMaxParts + h: its solver pool's
    static std::mutex mu;
    static hipStream_t pool[64][2 * kMaxParts] = {};
    int dev = 0;
    hipError_t err = hipGetDevice(&dev);
    if (err != hipSuccess) return err;
    if (dev < 0 || dev >= 64 || h < 0 || h >= 2 * kMaxParts) return hipErrorInvalidValue;
    std::lock_guard<std::mutex> lock(mu);
    for (int i = h < kMaxParts ? 0 : kMaxParts; i <= h; ++i)   // always in index order, so that stream i of a device is the i-th created of its kind
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
    bool fused;                 // cfg.reserved bit 4: k_tree_net drives the games (tree + narrow net in one kernel)
    bool pool_reset_pending;    // the batch was re-partitioned (raz_engine_set_parts): parked solver searches are re-dispatched first
    // The solver pool's round of a slice is launched after every pool_every-th tree launch of the slice.  pool_every == 1: on the
    // slice's stream, between the tree kernel and the net kernel (every step waits for the round).  pool_every > 1: on the slice's
    // POOL stream, beside the net kernel and the next pool_every - 1 tree launches of the slice - games that are not waiting for the
    // solver go on at the tree kernels' pace, the ones that are get an answer at the pool's - and joined before the launch that the
    // next round follows; raz_engine_step joins the last one before it returns, so every other entry point finds the pool at rest.
    int pool_every;
    hipStream_t pool_stream[kMaxParts];
    hipEvent_t ev_pool_go[kMaxParts], ev_pool_done[kMaxParts];
    bool pool_open[kMaxParts];      // a round of this slice is in flight on its pool stream
    uint32_t pool_tick[kMaxParts];  // tree launches of the slice since raz_engine_start
};

namespace {

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

// iterations of a worker lane per k_solve_run launch: raz_engine_config.reserved bits 16-23 x 64, or the default
inline int pool_budget_of(const raz_engine_dev& d) {
#ifdef RAZ_WAVE_EMU
    if (getenv("RAZ_SOLVER_BUDGET")) return atoi(getenv("RAZ_SOLVER_BUDGET"));   // (tests: park every search after a handful of iterations)
#endif
    const int units = (int)((d.cfg.reserved >> 16) & 0xffu);
    return units ? units * 64 : RAZ_SOLVER_POOL_BUDGET;
}

// The solver pool's round for slice h (raz_solver_pool.h): task trees for the requests the tree kernels have posted, `budget`
// iterations of the slice's share of the worker waves, answers for the scans that are decided.  On the slice's stream right after a
// tree launch, or on the slice's pool stream (pool_after_tree below).
int launch_solver_pool(raz_engine* e, int h, hipStream_t s) {
    const raz_engine_dev& d = e->dev;
    if (!d.W) return RAZ_OK;
    const Half hf = half_of(e, h);
    if (hf.count == 0) return RAZ_OK;
    const uint32_t wp = d.W / (uint32_t)e->parts;   // (W >= kMaxParts >= parts)
    hipLaunchKernelGGL(k_solve_scan, dim3(hf.count), dim3(64), 0, s, d, hf.g0, hf.count, (uint32_t)h, 1u);
    hipLaunchKernelGGL(k_solve_run, dim3(wp), dim3(64), 0, s, d, (uint32_t)h, wp * (uint32_t)h, wp, hf.g0, pool_budget_of(d));
    hipLaunchKernelGGL(k_solve_scan, dim3(hf.count), dim3(64), 0, s, d, hf.g0, hf.count, (uint32_t)h, 0u);
    return raz_check_launch("raz_engine_step: solver pool");
}

// pool_every > 1 (see raz_engine): before tree launch number `tick` of slice h on stream s - the round in flight is joined if this is
// the launch the next round follows ...
int pool_before_tree(raz_engine* e, int h, hipStream_t s) {
    if (e->pool_every <= 1 || !e->dev.W) return RAZ_OK;
    if (e->pool_open[h] && e->pool_tick[h] % (uint32_t)e->pool_every == 0u) {
        RAZ_HIP_TRY(hipStreamWaitEvent(s, e->ev_pool_done[h], 0), "raz_engine_step: solver pool join");
        e->pool_open[h] = false;
    }
    return RAZ_OK;
}
// ... and after it: the round starts on the pool stream once the tree kernel has posted its requests
int pool_after_tree(raz_engine* e, int h, hipStream_t s) {
    if (!e->dev.W) return RAZ_OK;
    if (e->pool_every <= 1) return launch_solver_pool(e, h, s);
    const uint32_t tick = e->pool_tick[h]++;
    if (tick % (uint32_t)e->pool_every != 0u) return RAZ_OK;
    if (!e->pool_stream[h]) {
        hipError_t err = raz_pool_stream(kMaxParts + h, &e->pool_stream[h]);
        if (err == hipSuccess) err = hipEventCreateWithFlags(&e->ev_pool_go[h], hipEventDisableTiming);
        if (err == hipSuccess) err = hipEventCreateWithFlags(&e->ev_pool_done[h], hipEventDisableTiming);
        if (err != hipSuccess) return raz_fail_hip(err, "raz_engine_step: solver pool stream");
    }
    RAZ_HIP_TRY(hipEventRecord(e->ev_pool_go[h], s), "raz_engine_step: solver pool fork");
    RAZ_HIP_TRY(hipStreamWaitEvent(e->pool_stream[h], e->ev_pool_go[h], 0), "raz_engine_step: solver pool fork");
    const int rc = launch_solver_pool(e, h, e->pool_stream[h]);
    if (rc != RAZ_OK) return rc;
    RAZ_HIP_TRY(hipEventRecord(e->ev_pool_done[h], e->pool_stream[h]), "raz_engine_step: solver pool done");
    e->pool_open[h] = true;
    return RAZ_OK;
}
// at the end of raz_engine_step: slice h's stream s waits for the round in flight
int pool_join(raz_engine* e, int h, hipStream_t s) {
    if (!e->pool_open[h]) return RAZ_OK;
    RAZ_HIP_TRY(hipStreamWaitEvent(s, e->ev_pool_done[h], 0), "raz_engine_step: solver pool join");
    e->pool_open[h] = false;
    return RAZ_OK;
}

// one simulation step of one slice on stream s; ev (nullable) = 3 events bracketing the two kernels
int launch_half_step(raz_engine* e, int h, hipStream_t s, hipEvent_t* ev) {
    const raz_engine_dev& d = e->dev;
    const Half hf = half_of(e, h);
    if (hf.count == 0) return RAZ_OK;
    const bool solver = d.cfg.use_solver_turn || d.cfg.use_solver_turn_in_simulation;
    if (solver) {
        const int rp = pool_before_tree(e, h, s);
        if (rp != RAZ_OK) return rp;
    }
    if (ev) hipEventRecord(ev[0], s);
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
    if (solver && (rc = pool_after_tree(e, h, s)) != RAZ_OK) return rc;
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

// k_tree_net (raz_engine_fused.hip): `n_steps` simulation steps of the whole batch
int launch_fused_steps(raz_engine* e, uint32_t n_steps, hipStream_t s) {
    const bool solver = e->dev.cfg.use_solver_turn || e->dev.cfg.use_solver_turn_in_simulation;
    if (!solver) return raz_launch_tree_net(e->dev, false, n_steps, (const float*)e->net.d_weights, e->net.res_layers, e->net.value_fc, s);
    // a game that posts a solve leaves its launch, so the solver pool gets its round after every launch of <= kFusedSolverIters steps
    constexpr uint32_t kFusedSolverIters = 8;
    int rc = RAZ_OK;
    while (n_steps && rc == RAZ_OK) {
        const uint32_t it = n_steps < kFusedSolverIters ? n_steps : kFusedSolverIters;
        rc = pool_before_tree(e, 0, s);
        if (rc == RAZ_OK) rc = raz_launch_tree_net(e->dev, true, it, (const float*)e->net.d_weights, e->net.res_layers, e->net.value_fc, s);
        if (rc == RAZ_OK) rc = pool_after_tree(e, 0, s);
        n_steps -= it;
    }
    const int rj = pool_join(e, 0, s);
    return rc != RAZ_OK ? rc : rj;
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
    e->pool_reset_pending = false;
    memset(&e->cache, 0, sizeof e->cache);
    e->cache_step = 1;
    // reserved bit 1: single stream; bits 8..11: number of slices (default 3); bits 12..15: kInnerMax override
    int parts = (int)((cfg->reserved >> 8) & 0xf);
    if (parts == 0) parts = 3;
    if (parts > kMaxParts) parts = kMaxParts;
    if (cfg->n_games < 256 || (cfg->reserved & 2u)) parts = 1;
    e->parts = parts;
    e->fused = (cfg->reserved & 16u) != 0;      // reserved bit 4: tree + narrow net in one kernel (k_tree_net)
    if (e->fused) {
        if (net->filters != 16 || net->value_fc > 1024) {
            delete e;
            return raz_fail(RAZ_EINVAL, "raz_engine_create: the fused tree + net kernels (reserved bit 4) need a 16-filter net");
        }
        parts = 1;
        e->parts = 1;
    }
    e->ev_fork = nullptr;
    e->pool_every = (int)((cfg->reserved >> 24) & 0xfu) ? (int)((cfg->reserved >> 24) & 0xfu) : RAZ_SOLVER_POOL_EVERY;
    e->dev.xk = e->pool_every > 1 ? 1u : 0u;   // tree and pool kernels run concurrently: their hand-off words are agent-scope (raz_engine_core.h xk_*)
    for (int h = 0; h < kMaxParts; ++h) {
        e->aux[h] = nullptr;
        e->ev_join[h] = nullptr;
        e->pool_stream[h] = nullptr;
        e->ev_pool_go[h] = e->ev_pool_done[h] = nullptr;
        e->pool_open[h] = false;
        e->pool_tick[h] = 0u;
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
    if (e->fused) return RAZ_OK;   // one kernel steps the whole batch
    if (e->dev.B < 256) parts = 1;
    hipError_t err = hipSuccess;
    if (parts > 1 && !e->ev_fork) err = hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming);
    for (int h = 1; h < parts && err == hipSuccess; ++h) {
        if (!e->aux[h]) err = raz_pool_stream(h, &e->aux[h]);
        if (err == hipSuccess && !e->ev_join[h]) err = hipEventCreateWithFlags(&e->ev_join[h], hipEventDisableTiming);
    }
    if (err != hipSuccess) return raz_fail_hip(err, "raz_engine_set_parts");
    if (parts != e->parts && e->dev.W) e->pool_reset_pending = true;   // worker waves change slices: raz_engine_step re-dispatches the parked searches
    e->parts = parts;
    return RAZ_OK;
}

extern "C" int raz_engine_set_solver_pool_every(raz_engine* e, int n) {
    if (!e || n < 0 || n > 15) return raz_fail(RAZ_EINVAL, "raz_engine_set_solver_pool_every: n must be 0..15");
    e->pool_every = n ? n : RAZ_SOLVER_POOL_EVERY;   // (raz_engine_step has joined the last round: the pool is at rest)
    e->dev.xk = e->pool_every > 1 ? 1u : 0u;
    for (int h = 0; h < kMaxParts; ++h) e->pool_tick[h] = 0u;
    return RAZ_OK;
}

extern "C" void raz_engine_destroy(raz_engine* e) {
    if (!e) return;
    for (int h = 0; h < kMaxParts; ++h) {
        // (aux streams belong to the process-wide pool)
        if (e->ev_join[h]) hipEventDestroy(e->ev_join[h]);
        if (e->ev_pool_go[h]) hipEventDestroy(e->ev_pool_go[h]);
        if (e->ev_pool_done[h]) hipEventDestroy(e->ev_pool_done[h]);
    }
    if (e->ev_fork) hipEventDestroy(e->ev_fork);
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
    if (d.M) RAZ_HIP_TRY(hipMemsetAsync(d.solver_ws, 0, (size_t)d.B * RAZ_SOLVER_WS_BYTES, s), "raz_engine_start: clear solver state");
    if (d.W) {
        RAZ_HIP_TRY(hipMemsetAsync(d.pool_hdr, 0, kPoolParts * sizeof(raz_solver_pool_hdr), s), "raz_engine_start: clear solver pool");
        RAZ_HIP_TRY(hipMemsetAsync(d.pool_state, 0, (size_t)d.W * RAZ_SOLVER_WORKER_STATE_BYTES, s), "raz_engine_start: clear solver pool");
        e->pool_reset_pending = false;
    }
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
int launch_steps_direct(raz_engine* e, uint32_t n_steps, hipStream_t s) {
    if (e->pool_reset_pending) {
        const size_t n = (size_t)e->dev.W * 64 > e->dev.B ? (size_t)e->dev.W * 64 : e->dev.B;
        hipLaunchKernelGGL(k_solve_pool_reset, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, e->dev);
        e->pool_reset_pending = false;
    }
    if (e->fused) return launch_fused_steps(e, n_steps, s);
    int rc = fork_aux(e, s);
    for (uint32_t i = 0; i < n_steps && rc == RAZ_OK; ++i) {
        for (int h = 0; h < e->parts && rc == RAZ_OK; ++h) rc = launch_half_step(e, h, stream_of(e, h, s), nullptr);
    }
    for (int h = 0; h < kMaxParts; ++h) {
        const int rp = pool_join(e, h, stream_of(e, h, s));
        if (rc == RAZ_OK) rc = rp;
    }
    const int rj = join_aux(e, s);
    return rc != RAZ_OK ? rc : rj;
}
}  // namespace

// (A hipGraph replay of 16 steps x slices x 2 kernels was measured 6 % slower than these direct launches on ROCm 7.2 / MI355X in
// rounds 1-3 and removed in round 4: DESIGN.md section 4, "tried".)
extern "C" int raz_engine_step(raz_engine* e, uint32_t n_steps, raz_stream_t stream) {
    if (!e) return raz_fail(RAZ_EINVAL, "raz_engine_step: NULL engine");
    if (!e->started) return raz_fail(RAZ_ESTATE, "raz_engine_step: call raz_engine_start first");
    return launch_steps_direct(e, n_steps, (hipStream_t)stream);
}

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
    if (e->fused) {   // one kernel does both: its time is reported as the tree kernel's
        hipEvent_t a, b;
        RAZ_HIP_TRY(hipEventCreate(&a), "raz_engine_step_timed: hipEventCreate");
        RAZ_HIP_TRY(hipEventCreate(&b), "raz_engine_step_timed: hipEventCreate");
        hipEventRecord(a, s);
        const int rc = launch_fused_steps(e, n_steps, s);
        hipEventRecord(b, s);
        const hipError_t err = hipStreamSynchronize(s);
        float ms = 0.f;
        if (rc == RAZ_OK && err == hipSuccess) hipEventElapsedTime(&ms, a, b);
        *tree_ms += ms;
        hipEventDestroy(a);
        hipEventDestroy(b);
        if (err != hipSuccess) return raz_fail_hip(err, "raz_engine_step_timed: sync");
        return rc;
    }
    const int H = e->parts;
    std::vector<hipEvent_t> ev(3 * (size_t)n_steps * H);
    for (auto& x : ev) RAZ_HIP_TRY(hipEventCreate(&x), "raz_engine_step_timed: hipEventCreate");
    int rc = fork_aux(e, s);
    for (uint32_t i = 0; i < n_steps && rc == RAZ_OK; ++i) {
        for (int h = 0; h < H && rc == RAZ_OK; ++h)
            rc = launch_half_step(e, h, stream_of(e, h, s), &ev[3 * ((size_t)i * H + h)]);
    }
    for (int h = 0; h < kMaxParts; ++h) {
        const int rp = pool_join(e, h, stream_of(e, h, s));
        if (rc == RAZ_OK) rc = rp;
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
    if (e->fused && d_cache) return raz_fail(RAZ_EINVAL, "raz_engine_set_leaf_cache: the fused tree + net kernel evaluates leaves in place (no leaf exchange to cache)");
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

extern "C" int raz_engine_solver_stats(raz_engine* e, uint64_t* out12, raz_stream_t stream) {
    if (!e || !out12) return raz_fail(RAZ_EINVAL, "raz_engine_solver_stats: NULL argument");
    static_assert(sizeof(unsigned long long) == sizeof(uint64_t), "counter width");
    if (e->dev.W) hipLaunchKernelGGL(k_solver_game_stats, dim3(1), dim3(256), 0, (hipStream_t)stream, e->dev);
    RAZ_HIP_TRY(hipMemcpyAsync(out12, e->dev.counters + 17, 15 * sizeof(uint64_t), hipMemcpyDeviceToHost, (hipStream_t)stream), "raz_engine_solver_stats: copy");
    RAZ_HIP_TRY(hipStreamSynchronize((hipStream_t)stream), "raz_engine_solver_stats: sync");
    return RAZ_OK;
}

extern "C" int raz_engine_set_resign_threshold(raz_engine* e, int has_threshold, double threshold) {
    if (!e) return raz_fail(RAZ_EINVAL, "raz_engine_set_resign_threshold: NULL engine");
    e->dev.cfg.has_resign_threshold = has_threshold ? 1 : 0;
    e->dev.cfg.resign_threshold = threshold;
    return RAZ_OK;
}

// diagnostics: `bytes` at `offset` of one of raz_engine_device_ptr's arrays, copied to the host (synchronises the device)
extern "C" int raz_engine_debug_read(raz_engine* e, int which, size_t offset, size_t bytes, void* host_out);

// One of the engine's device arrays and its size in bytes (0: no such array in this engine).
static unsigned char* device_array(raz_engine* e, int which, size_t* bytes) {
    size_t n = 0;
    void* p = nullptr;
    if (e) {
        const raz_engine_dev& d = e->dev;
        const size_t B = d.B, MP = d.max_plies, BK = (size_t)d.B * d.K;
        switch (which) {
            case 0: p = d.rec; n = B * MP * sizeof(raz_ply_header); break;
            case 1: p = d.rec_n; n = B * MP * 64 * 4; break;
            case 2: p = d.rec_w; n = d.rec_w ? B * MP * 64 * 8 : 0; break;
            case 3: p = d.game; n = B * sizeof(raz_game); break;
            case 5: p = d.prof; n = B * 8 * 8; break;
            case 6: p = d.solver_ws; n = d.solver_ws ? B * (size_t)RAZ_SOLVER_WS_BYTES : 0; break;   // diagnostics of the solver pool (tools/sessions/debug_solver_stall.py)
            case 7: p = d.pool_state; n = (size_t)d.W * RAZ_SOLVER_WORKER_STATE_BYTES; break;
            case 8: p = d.pool_hdr; n = d.W ? kMaxParts * sizeof(raz_solver_pool_hdr) : 0; break;
            case 9: p = d.pool_active; n = d.W ? B * 4 : 0; break;
            // the leaf exchange between the tree kernels and the net (one row per simulation slot, row = game * K + slot): what the
            // LAST step asked the net and what it answered (bench.py's net_check reads the timed batch's rows)
            case 10: p = d.nn_active; n = BK; break;
            case 11: p = d.nn_own; n = BK * 8; break;
            case 12: p = d.nn_enemy; n = BK * 8; break;
            case 13: p = d.nn_policy; n = BK * 64 * 4; break;
            case 14: p = d.nn_value; n = BK * 4; break;
            default: break;
        }
    }
    if (bytes) *bytes = p ? n : 0;
    return n ? (unsigned char*)p : nullptr;
}

extern "C" void* raz_engine_device_ptr(raz_engine* e, int which) { return device_array(e, which, nullptr); }

extern "C" int raz_engine_debug_read(raz_engine* e, int which, size_t offset, size_t bytes, void* host_out) {
    size_t size = 0;
    unsigned char* p = device_array(e, which, &size);
    if (!p || !host_out) return raz_fail(RAZ_EINVAL, "raz_engine_debug_read: no such array");
    if (offset > size || bytes > size - offset) return raz_fail(RAZ_EINVAL, "raz_engine_debug_read: offset + bytes beyond the array");
    RAZ_HIP_TRY(hipDeviceSynchronize(), "raz_engine_debug_read: sync");
    RAZ_HIP_TRY(hipMemcpy(host_out, p + offset, bytes, hipMemcpyDeviceToHost), "raz_engine_debug_read: copy");
    return RAZ_OK;
}
