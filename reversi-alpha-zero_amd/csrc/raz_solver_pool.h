// raz_solver_pool.h — the end-game solver as a POOL OF WORKER LANES shared by all games (lib/alt/reversi_solver_cython.pyx:40-127;
// call sites agent/player.py:100-103,150-161 - exact, at the root - and :237-251 - win/loss, inside simulations).
//
// Round 4 searched a game's position inside the game's own wave: the top three plies became <= 2184 TASKS (subtrees), the wave's 64
// lanes ran them.  A win/loss solve needs only the few tasks the reference's early-stopping scan really visits, so on mini.yml as
// shipped 13 of a wave's 64 lanes executed an instruction (profiles/r5_pmc/solver_bound_*three_ply_build*: 20 %; 12.7 % with two-ply
// tasks) - and every game held a whole wave (and 248 registers of the tree kernel) for it.  Here the tree kernels only POST the
// position (raz_engine_core.h solver_solve) and three launches between two tree launches do the work:
//   k_solve_scan (one wave per game that has a request)  builds the task tree of a new request in the game's block of E.solver_ws -
//       child i, its j-th move, the m-th move after that (three plies, folded in LDS), and below every such position that still has
//       >= RAZ_SOLVER_SPLIT_EMPTIES empties a FOURTH ply: its moves are the tasks (SolverDeep, in HBM) - and lists the game as ACTIVE;
//   k_solve_run  (W worker waves that belong to no game)  every LANE draws a task - lane L of wave w walks its own sequence over the
//       active list, a solve's tasks leave in the reference's scan order by the solve's own counter, so the lanes go first to the
//       tasks the sequential search needs - and runs the reference's depth-first search on its subtree: current node in registers,
//       ancestors' frames in LDS (in the pool's arrays in HBM between launches), one move per iteration with the returns of finished
//       nodes folded in; memo traffic and task draws only in a slow phase every RAZ_SOLVER_SLOW_EVERY-th iteration.  After `budget`
//       iterations the lanes park their searches; the next launch picks them up;
//   k_solve_scan again  folds the finished tasks into level-3 values, those into level-2 values, those into the root's children,
//       those into the root's answer - each scan the reference's loop over a node's moves in ascending order, strict improvement,
//       win/loss mode stopping at the first value > 0 - and publishes an answer as soon as the root's scan is decided; tasks behind a
//       decided node are never started (one dead-flag byte per level-3 node), searches of them are dropped at the next launch.
// One round = scan, run, scan.  It follows a tree launch on the slice's stream, or - raz_engine.hip, pool_every > 1 - runs on a stream of
// its own beside the next tree launches (games that wait for no solve are not held up by it; continuous batching: 22.7 M -> 30.3 M
// sims/s on mini.yml as shipped): an answer is ONE 4-byte store of the header's state word (raz_engine.h), a request is complete
// before its state says so, and a memo entry whose tag is visible before its keys is a miss.
// f is a function of (position, mode): which lane searches a subtree, in which launch, or whether a subtree the sequential scan
// would have skipped is searched as well changes no answer.  Workers never talk to each other inside a launch (only the solves' task
// counters and the claim of a memo slot are atomic); everything else crosses kernel boundaries.  Measurements: DESIGN.md 4.7.
#pragma once
#include "raz_engine_core.h"
#include "raz_bitboard_valu.h"   // the per-LANE forms of the bitboard primitives (same results for every input)

namespace {

#define RAZ_SOLVER_MAX_L2 (RAZ_SOLVER_MAX_DEPTH * (RAZ_SOLVER_MAX_DEPTH - 1))        // 182 positions two plies below the root
#define RAZ_SOLVER_MAX_TASKS (RAZ_SOLVER_MAX_L2 * (RAZ_SOLVER_MAX_DEPTH - 2))        // 2184 subtrees three plies below it
#define RAZ_SOLVER_UNKNOWN (-128)
struct SolverTree {   // in the game's block of E.solver_ws behind the header; k_solve_scan works on a copy in LDS
    unsigned long long c_own[RAZ_SOLVER_MAX_DEPTH], c_enemy[RAZ_SOLVER_MAX_DEPTH], c_moves[RAZ_SOLVER_MAX_DEPTH];   // child i: position (its mover's view), its moves
    unsigned long long g_own[RAZ_SOLVER_MAX_L2], g_enemy[RAZ_SOLVER_MAX_L2], g_moves[RAZ_SOLVER_MAX_L2];           // level-2 node n, likewise
    unsigned short g_first[RAZ_SOLVER_MAX_L2 + 2];   // node n's first task
    unsigned char c_first[RAZ_SOLVER_MAX_DEPTH + 2]; // child i's first level-2 node
    signed char result[RAZ_SOLVER_MAX_TASKS];        // level-3 node t: the value of the move that leads to it, for the level-2 node's mover (RAZ_SOLVER_UNKNOWN: not there yet)
    signed char c_v[RAZ_SOLVER_MAX_DEPTH];           // child i: the value of the root's i-th move once known, for the ROOT's mover (else RAZ_SOLVER_UNKNOWN)
    signed char g_v[RAZ_SOLVER_MAX_L2];              // node n: the value of the reply that leads to it once known, for the CHILD's mover
    unsigned char c_kind[RAZ_SOLVER_MAX_DEPTH];      // child i: 0 the game ends there, 1 the opponent moves, 2 the opponent passes (+4: from the memo)
    unsigned char g_kind[RAZ_SOLVER_MAX_L2];         // node n, likewise (seen from the child's mover)
    unsigned char c_a[RAZ_SOLVER_MAX_DEPTH];         // the square of the root's i-th move
    unsigned char g_child[RAZ_SOLVER_MAX_L2];        // node n's child
    unsigned short task_entry[RAZ_SOLVER_MAX_TASKS]; // level-3 node t: its level-2 node | that node's child << 8 (one request tells a worker both)
};
// Below the three plies of SolverTree: level-3 node t = the position after level-2 node n's j-th move (what round 4 and the first
// pool handed to a lane whole).  One subtree of a 10-empties solve is then up to 5040 leaf paths, and a solve is as slow as the
// longest subtree its scan needs - which set the pace of the whole lock-step batch (profiles/r5: the slowest game listed for 1492
// rounds of the pool, 3.7 per solve, whatever the number of lanes).  So a level-3 node with >= RAZ_SOLVER_SPLIT_EMPTIES empties is
// split once more: its moves are the tasks (a FOURTH ply), folded by k_solve_scan with the same scan; smaller ones are one task.
#define RAZ_SOLVER_SPLIT_EMPTIES 7
#define RAZ_SOLVER_MAX_SUB (RAZ_SOLVER_MAX_TASKS * (RAZ_SOLVER_MAX_DEPTH - 3))   // 24024
struct SolverDeep {   // HBM only (behind the SolverTree in the game's block)
    unsigned long long h_own[RAZ_SOLVER_MAX_TASKS], h_enemy[RAZ_SOLVER_MAX_TASKS], h_moves[RAZ_SOLVER_MAX_TASKS];   // level-3 node t (its mover's view), its moves
    unsigned short sub_first[RAZ_SOLVER_MAX_TASKS + 4];   // node t's first task
    unsigned char h_dead[RAZ_SOLVER_MAX_TASKS];           // node t: its value is known or no longer needed (a scan above it is decided): its tasks are moot
    unsigned char h_kind[RAZ_SOLVER_MAX_TASKS];           // node t: 0 the game ends there, 1 the opponent moves, 2 the opponent passes (+4: from the memo; +8: not split - ONE task, the node itself)
    signed char sub_result[RAZ_SOLVER_MAX_SUB];           // task u: the value of that move for node t's mover (an unsplit node: f of the node)
    unsigned short sub_task[RAZ_SOLVER_MAX_SUB];          // task u's level-3 node
};
static_assert(sizeof(SolverDeep) <= RAZ_SOLVER_DEEP_BYTES, "SolverDeep must fit the game's solver block");
static_assert(sizeof(SolverTree) <= RAZ_SOLVER_TREE_BYTES, "SolverTree must fit the game's solver block");
static_assert(sizeof(SolverTree) % 8 == 0, "SolverTree is copied in 8-byte words");

__device__ __forceinline__ SolverTree* solve_tree(const raz_engine_dev& E, uint32_t g) {
    return (SolverTree*)(E.solver_ws + (size_t)g * RAZ_SOLVER_WS_BYTES + sizeof(raz_solve_hdr));
}
__device__ __forceinline__ SolverDeep* solve_deep(const raz_engine_dev& E, uint32_t g) {
    return (SolverDeep*)(E.solver_ws + (size_t)g * RAZ_SOLVER_WS_BYTES + sizeof(raz_solve_hdr) + RAZ_SOLVER_TREE_BYTES);
}

// the reference's loop over a node's moves, on values that are already there: `vals` in ascending move order, RAZ_SOLVER_UNKNOWN =
// not there yet.  Returns false while the scan is not decided.  Non-exact: it ends at the first value > 0
__device__ __forceinline__ bool solver_scan(const signed char* vals, int n, raz_bb moves, bool exact, int& bm, int& bs) {
    bm = -1;
    bs = -100;
    raz_bb m = moves;
    for (int j = 0; j < n; ++j, m &= m - 1) {
        const int v = vals[j];
        if (v == RAZ_SOLVER_UNKNOWN) return false;
        if (bs < v) {
            bm = __ffsll((long long)m) - 1;
            bs = v;
        }
        if (!exact && bs > 0) break;
    }
    return true;
}

// a position after `mover` (own, enemy) played square a: who moves next.  kind 0: the game ends (v = disc difference for `mover`);
// 1: the opponent moves; 2: the opponent passes (the mover again); (no, ne, nm) = the next position from ITS mover's view and its moves
// RAZ_SOLVER_INLINE_LAST: a position with ONE empty square left is not handed back as a node - it would cost the worker wave a whole
// iteration to play its only move - but finished here: whoever can play the square plays it (one more calc_flip), the discs are counted,
// and the move reports "the game ends there" with that count.  The same value the two steps give (a node with one move has nothing to
// choose and nothing to cut off); in a full-width search every second node is such a node.
#ifndef RAZ_SOLVER_INLINE_LAST
#define RAZ_SOLVER_INLINE_LAST 1   // 2: positions with TWO empty squares are finished in the move function too (solver_last_two)
// Measured on mini.yml as shipped (M sims/s: two-kernel lock-step / continuous batching / fused lock-step; tools/sessions/r6_s22-24.sh,
// profiles/r6/solver_last_squares_finished_in_the_move_function_ab.json), level x iterations per round:
//   0 x 128: 31.2 / 31.4 / 25.4 (4.61 rounds per answer)     1 x 128: 33.7 / 32.6 / 29.5 (3.28)     1 x 96: 34.2 / 33.7 / 27.5     1 x 64: 33.6 / 32.9 / 23.7
//   2 x 128: 32.0 / 30.6 / 32.1 (2.15: fewer, longer iterations - a round of 128 takes too long)     2 x 96: 33.2 / 31.6 / 30.4     2 x 80: 33.8 / 32.5 / 29.5     2 x 64: 33.8 / 33.3 / 27.7
// Level 1 at 96 iterations per round is the default (the worker runs the two-kernel pipeline with continuous batching when the solver is on).
#endif
// one empty square e, `own` to move: the final disc difference for `own` (env/reversi_env.py:68-85: the mover plays it if that flips
// something, else the opponent does, else the game is over as it stands)
__device__ __forceinline__ int solver_last_one(int e, raz_bb own, raz_bb enemy) {
    const int f = bb_popcount(bbv_calc_flip(e, own, enemy));
    const int po = bb_popcount(own), pe = bb_popcount(enemy);
    if (f) return (po + f + 1) - (pe - f);
    const int g = bb_popcount(bbv_calc_flip(e, enemy, own));
    return g ? (po - g) - (pe + g + 1) : po - pe;
}
// two empty squares, `own` to move with the legal moves `moves` (not empty): the reference's loop over them (ascending, strict
// improvement, non-exact: done at the first value > 0) with each reply finished by solver_last_one
__device__ __forceinline__ int solver_last_two(raz_bb moves, raz_bb own, raz_bb enemy, bool exact) {
    const raz_bb empties = ~(own | enemy);
    int bs = -100;
    for (raz_bb m = moves; m; m &= m - 1) {
        const int s = __ffsll((long long)m) - 1;
        const raz_bb fl = bbv_calc_flip(s, own, enemy);
        const raz_bb o2 = (own ^ fl) | (1ULL << s), e2 = enemy ^ fl;
        const int val = -solver_last_one(__ffsll((long long)(empties & ~(1ULL << s))) - 1, e2, o2);
        if (bs < val) bs = val;
        if (!exact && bs > 0) break;
    }
    return bs;
}
__device__ __forceinline__ int solver_play(int a, raz_bb own, raz_bb enemy, raz_bb& no, raz_bb& ne, raz_bb& nm, int& v, bool exact) {
    const raz_bb flipped = bbv_calc_flip(a, own, enemy);
    const raz_bb nown = (own ^ flipped) | (1ULL << a), nenemy = enemy ^ flipped;
    const raz_bb l1 = bbv_legal_moves(nenemy, nown);
    const raz_bb l2 = l1 ? 0ULL : bbv_legal_moves(nown, nenemy);
    if (!(l1 | l2)) {
        v = bb_popcount(nown) - bb_popcount(nenemy);
        no = ne = nm = 0ULL;
        return 0;
    }
    if (RAZ_SOLVER_INLINE_LAST >= 2 && bb_popcount(~(nown | nenemy)) == 2) {
        // two squares left: the node's mover is the opponent if it can move (l1), else the mover of this move again (l2)
        const bool opp = l1 != 0ULL;
        const int nv = solver_last_two(l1 | l2, opp ? nenemy : nown, opp ? nown : nenemy, exact);
        v = opp ? -nv : nv;
        no = ne = nm = 0ULL;
        return 0;
    }
    if (RAZ_SOLVER_INLINE_LAST && bb_popcount(~(nown | nenemy)) == 1) {
        // the last square: the opponent's if it can play it (l1), else the mover's again (l2)
        const bool opp = l1 != 0ULL;
        const raz_bb last_own = opp ? nenemy : nown, last_enemy = opp ? nown : nenemy;
        const int f = bb_popcount(bbv_calc_flip(__ffsll((long long)(l1 | l2)) - 1, last_own, last_enemy));
        const int last = bb_popcount(last_own) + f + 1, other = bb_popcount(last_enemy) - f;
        v = opp ? other - last : last - other;   // (for the mover of THIS move)
        no = ne = nm = 0ULL;
        return 0;
    }
    no = l1 ? nenemy : nown;
    ne = l1 ? nown : nenemy;
    nm = l1 ? l1 : l2;
    v = 0;
    return l1 ? 1 : 2;
}

#ifndef RAZ_SOLVER_DRAW_GATE
#define RAZ_SOLVER_DRAW_GATE 1   // idle lanes draw tasks in proportion to the tasks left (k_solve_run, "idle lanes draw tasks"); 0: every idle lane at every slow phase
#endif
#ifndef RAZ_SOLVER_NE_WINDOW
#define RAZ_SOLVER_NE_WINDOW 0   // open root moves of a win/loss solve whose tasks are handed out; 0: all of them at once (the default: see k_solve_scan)
#endif

// ------------------------------------------------------------------ k_solve_scan
// collect = 1 (before k_solve_run): new requests get their task tree, every solve with tasks left is listed as active.
// collect = 0 (after it): the workers' results are folded in; block 0 clears the list for the next round.
// Both passes run the same scans, so an answer is published by whichever pass first sees its last missing value.
__global__ __launch_bounds__(64) void k_solve_scan(raz_engine_dev E, uint32_t g0, uint32_t count, uint32_t part, uint32_t collect) {
    if (blockIdx.x >= count) return;
    const int lane = threadIdx.x;
    raz_solver_pool_hdr* ph = E.pool_hdr + part;
    if (!collect && blockIdx.x == 0 && lane == 0) ph->n_active = 0u;   // (nobody reads it between k_solve_run and the next collect pass)
    const uint32_t g = g0 + blockIdx.x;
    if (g >= E.B) return;
    raz_solve_hdr* h = solve_hdr(E, g);
    // (the tree kernels may be running beside this launch: a request's words are read the way they are published, raz_engine_core.h xk_*;
    // the position is requested after the state has arrived - a REQUESTED state is stored after its position has drained)
    const bool conc = E.xk != 0u;
    const uint32_t st = RAZ_SOLVE_STATE(uni(xk_load32(conc, &h->state)));
    if (st != RAZ_SOLVE_REQUESTED && st != RAZ_SOLVE_RUNNING) return;
    __shared__ SolverTree tree_lds;
    SolverTree* P = &tree_lds;
    SolverTree* T = solve_tree(E, g);
    SolverDeep* D = solve_deep(E, g);
    const raz_bb own0 = uni((raz_bb)xk_load64(conc, &h->own0)), enemy0 = uni((raz_bb)xk_load64(conc, &h->enemy0));
    const uint32_t exact = uni(xk_load32(conc, &h->exact));
    int k = 0, n2 = 0, total = 0, subs = 0;   // root moves, level-2 nodes, level-3 nodes, tasks
    if (st == RAZ_SOLVE_REQUESTED) {
        const raz_bb legal0 = bb_legal_moves(own0, enemy0);
        k = bb_popcount(legal0);
        if (k == 0) {   // the reference's (None, None): never the case for a running game
            if (lane == 0) {
                h->ans_move = -1;
                xk_store32(conc, &h->state, RAZ_SOLVE_ANSWER_WORD(RAZ_SOLVE_NONE, -1, -100));
            }
            return;
        }
        // ---- ply 1: lane i < k owns the root's i-th move.  kind 0: the game ends there (my_v = disc difference); 1: the opponent
        // moves; 2: the opponent passes; +4: f(child) came from the memo
        int my_a = -1, my_kind = 0, my_v = 0, my_nodes = 0, first = 0;
        raz_bb c_own = 0, c_enemy = 0, c_moves = 0;
        if (lane < k) {
            raz_bb m = legal0;
            for (int i = 0; i < lane; ++i) m &= m - 1;
            my_a = __ffsll((long long)m) - 1;
            my_kind = solver_play(my_a, own0, enemy0, c_own, c_enemy, c_moves, my_v, exact != 0u);
            if (my_kind) {
                int rm, rs;
                if (bb_popcount(~(c_own | c_enemy)) >= 4 && memo_find_lane(E, g, c_own, c_enemy, exact, rm, rs)) {
                    my_kind |= 4;
                    my_v = (my_kind & 1) ? -rs : rs;
                } else
                    my_nodes = bb_popcount(c_moves);
            }
        }
        for (int i = 0; i < k; ++i) {   // exclusive prefix sum over the lanes: child i's level-2 nodes are [first, first + my_nodes)
            const int ti = (int)lane_u32((uint32_t)my_nodes, i);
            if (i < lane) first += ti;
            n2 += ti;
        }
        wave_sync();
        if (lane < k) {
            P->c_own[lane] = c_own;
            P->c_enemy[lane] = c_enemy;
            P->c_moves[lane] = c_moves;
            P->c_first[lane] = (unsigned char)first;
            P->c_kind[lane] = (unsigned char)my_kind;
            P->c_a[lane] = (unsigned char)my_a;
            P->c_v[lane] = (signed char)((my_kind == 0 || (my_kind & 4)) ? my_v : RAZ_SOLVER_UNKNOWN);   // known now: the game ends there, or the memo had f(child)
        }
        if (lane == 0) P->c_first[k] = (unsigned char)n2;
        wave_sync_lanes();
        // ---- ply 2: level-2 node n = the position after child ci's j-th move, seen from the side to move there (kinds as at ply 1,
        // from the CHILD's mover's point of view).  Its moves are the tasks
        for (int n = lane; n < n2; n += 64) {
            int ci = 0;
            while (ci + 1 < k && (int)P->c_first[ci + 1] <= n) ++ci;
            raz_bb m = P->c_moves[ci];
            for (int j = (int)P->c_first[ci]; j < n; ++j) m &= m - 1;
            raz_bb go, ge, gm;
            int gv;
            int gk = solver_play(__ffsll((long long)m) - 1, P->c_own[ci], P->c_enemy[ci], go, ge, gm, gv, exact != 0u);
            int tasks = 0;
            if (gk) {
                int rm, rs;
                gv = RAZ_SOLVER_UNKNOWN;
                if (bb_popcount(~(go | ge)) >= 4 && memo_find_lane(E, g, go, ge, exact, rm, rs)) {
                    gk |= 4;
                    gv = (gk & 1) ? -rs : rs;
                } else
                    tasks = bb_popcount(gm);
            }
            P->g_own[n] = go;
            P->g_enemy[n] = ge;
            P->g_moves[n] = gm;
            P->g_kind[n] = (unsigned char)gk;
            P->g_v[n] = (signed char)gv;
            P->g_child[n] = (unsigned char)ci;
            P->g_first[n] = (unsigned short)tasks;   // (a count for now)
        }
        wave_sync_lanes();
        if (lane == 0) {   // counts -> first task of every node
            int acc = 0;
            for (int n = 0; n < n2; ++n) {
                const int c = P->g_first[n];
                P->g_first[n] = (unsigned short)acc;
                acc += c;
            }
            P->g_first[n2] = (unsigned short)acc;
        }
        wave_sync_lanes();
        total = (int)uni((uint32_t)P->g_first[n2]);
        for (int t = lane; t < total; t += 64) P->result[t] = (signed char)RAZ_SOLVER_UNKNOWN;
        for (int n = lane; n < n2; n += 64)
            for (int t = P->g_first[n]; t < (int)P->g_first[n + 1]; ++t) P->task_entry[t] = (unsigned short)(n | ((int)P->g_child[n] << 8));
        wave_sync_lanes();
        // ---- ply 3: level-3 node t = the position after level-2 node n's j-th move (lane l takes a contiguous run of them, so that
        // the tasks' numbering follows the scan order).  The game may end there, the memo may know it; else it is one task, or - with
        // >= RAZ_SOLVER_SPLIT_EMPTIES empties - as many as it has moves
        {
            const int per = (total + 63) / 64;
            const int t0 = lane * per < total ? lane * per : total, t1 = t0 + per < total ? t0 + per : total;
            int mine = 0;
            for (int t = t0; t < t1; ++t) {
                const int n = P->task_entry[t] & 0xff;
                raz_bb m = P->g_moves[n];
                for (int j = (int)P->g_first[n]; j < t; ++j) m &= m - 1;
                raz_bb no, ne, nm;
                int v, cnt = 0;
                int kind = solver_play(__ffsll((long long)m) - 1, P->g_own[n], P->g_enemy[n], no, ne, nm, v, exact != 0u);
                if (!kind)
                    P->result[t] = (signed char)v;
                else {
                    int rm, rs;
                    const int emp = bb_popcount(~(no | ne));
                    if (emp >= 4 && memo_find_lane(E, g, no, ne, exact, rm, rs)) {
                        kind |= 4;
                        P->result[t] = (signed char)((kind & 1) ? -rs : rs);
                    } else if (emp >= RAZ_SOLVER_SPLIT_EMPTIES)
                        cnt = bb_popcount(nm);
                    else {
                        kind |= 8;
                        cnt = 1;
                    }
                }
                D->h_own[t] = no;
                D->h_enemy[t] = ne;
                D->h_moves[t] = nm;
                D->h_kind[t] = (unsigned char)kind;
                D->h_dead[t] = 0;
                D->sub_first[t] = (unsigned short)cnt;   // (a count for now)
                mine += cnt;
            }
            int before = 0;
            for (int i = 0; i < 64; ++i) {
                const int x = (int)lane_u32((uint32_t)mine, i);
                if (i < lane) before += x;
                subs += x;
            }
            for (int t = t0; t < t1; ++t) {
                const int c = D->sub_first[t];
                D->sub_first[t] = (unsigned short)before;
                for (int u = before; u < before + c; ++u) {
                    D->sub_task[u] = (unsigned short)t;
                    D->sub_result[u] = (signed char)RAZ_SOLVER_UNKNOWN;
                }
                before += c;
            }
            if (lane == 0) D->sub_first[total] = (unsigned short)subs;
        }
    } else {
        k = (int)(uni(h->k_n2) & 0xffu);
        n2 = (int)(uni(h->k_n2) >> 8);
        total = (int)uni(h->tasks);
        subs = (int)uni(h->total);
        for (int i = lane; i < (int)(sizeof(SolverTree) / 8); i += 64) ((unsigned long long*)P)[i] = ((const unsigned long long*)T)[i];
        wave_sync_lanes();
        // ---- level-3 nodes: the tasks' values fold into result[] (the reference's loop over the node's moves again)
        for (int t = lane; t < total; t += 64) {
            if (P->result[t] != RAZ_SOLVER_UNKNOWN) continue;
            const int kind = D->h_kind[t], s0 = D->sub_first[t], s1 = D->sub_first[t + 1];
            int bm = -1, bs = 0;
            bool ok;
            if (kind & 8) {   // (its one task searched the node itself and remembered it)
                bs = D->sub_result[s0];
                ok = bs != RAZ_SOLVER_UNKNOWN;
            } else
                ok = solver_scan(D->sub_result + s0, s1 - s0, D->h_moves[t], exact != 0, bm, bs);
            if (ok) {
                if (!(kind & 8)) {
                    const raz_bb ho = D->h_own[t], he = D->h_enemy[t];
                    if (bb_popcount(~(ho | he)) >= 4) memo_put_lane(E, g, ho, he, exact, bm, bs);
                }
                const signed char r = (signed char)((kind & 1) ? -bs : bs);
                P->result[t] = r;
                T->result[t] = r;   // (the workers look at it before they start or go on with a task of this node)
            }
        }
    }
    wave_sync_lanes();
    // ---- level-2 nodes: the reference's loop over the node's moves, on the results that are there.  A value is f of that position
    // whoever asked for it, so it goes to the memo the moment it is known
    for (int n = lane; n < n2; n += 64) {
        const int gk = P->g_kind[n];
        if (P->g_v[n] == RAZ_SOLVER_UNKNOWN && (gk & 3) && !(gk & 4)) {
            const int t0 = P->g_first[n];
            int bm, bs;
            if (solver_scan(P->result + t0, (int)P->g_first[n + 1] - t0, P->g_moves[n], exact != 0, bm, bs)) {
                const raz_bb go = P->g_own[n], ge = P->g_enemy[n];
                if (bb_popcount(~(go | ge)) >= 4) memo_put_lane(E, g, go, ge, exact, bm, bs);
                P->g_v[n] = (signed char)((gk & 1) ? -bs : bs);
            }
        }
    }
    wave_sync_lanes();
    // ---- the root's children: lane i scans child i's level-2 nodes the same way
    if (lane < k) {
        const int ck = P->c_kind[lane];
        if (P->c_v[lane] == RAZ_SOLVER_UNKNOWN && (ck & 3) && !(ck & 4)) {
            const int n0 = P->c_first[lane];
            int bm, bs;
            if (solver_scan(P->g_v + n0, (int)P->c_first[lane + 1] - n0, P->c_moves[lane], exact != 0, bm, bs)) {
                const raz_bb co = P->c_own[lane], ce = P->c_enemy[lane];
                if (bb_popcount(~(co | ce)) >= 4) memo_put_lane(E, g, co, ce, exact, bm, bs);
                P->c_v[lane] = (signed char)((ck & 1) ? -bs : bs);
            }
        }
    }
    wave_sync_lanes();
    // ---- the root (every lane runs the same scan over LDS: the outcome is wave-uniform)
    int bm = -1, bs = -100;
    bool decided = true;
    for (int i = 0; i < k; ++i) {
        const int v = P->c_v[i];
        if (v == RAZ_SOLVER_UNKNOWN) {
            decided = false;
            break;
        }
        if (bs < v) {
            bm = P->c_a[i];
            bs = v;
        }
        if (!exact && bs > 0) break;
    }
    decided = uni((uint32_t)decided) != 0u;
    if (decided) {
        bm = uni(bm);
        bs = uni(bs);
        wave_sync();
        memo_put(E, g, own0, enemy0, exact, bm, bs, lane);
        if (lane == 0) {
            h->ans_move = bm;
            // ONE agent-scope 4-byte store (write-through): a tree kernel running beside this launch on another XCD finds the answer at
            // its next look instead of at the next kernel boundary (a plain store stayed in this XCD's L2 until the launch ended)
            xk_store32(conc, &h->state, RAZ_SOLVE_ANSWER_WORD(bm >= 0 ? RAZ_SOLVE_DONE : RAZ_SOLVE_NONE, bm, bs));
            atomicAdd(&E.counters[21], 1ULL);
            atomicAdd(&E.counters[22], (unsigned long long)(st == RAZ_SOLVE_REQUESTED ? 0u : h->rounds));
        }
        return;
    }
    // ---- not decided yet: what the workers look at goes (back) to the game's block
    if (st == RAZ_SOLVE_REQUESTED) {
        for (int i = lane; i < (int)(sizeof(SolverTree) / 8); i += 64) ((unsigned long long*)T)[i] = ((const unsigned long long*)P)[i];
        if (lane == 0) {
            h->rounds = 0u;
            h->k_n2 = (uint32_t)k | ((uint32_t)n2 << 8);
            h->tasks = (uint32_t)total;
            h->total = (uint32_t)subs;
            h->next = 0u;
            xk_store32(conc, &h->state, RAZ_SOLVE_RUNNING);   // (the tree kernels treat REQUESTED and RUNNING alike: nothing of the task tree is theirs to read)
        }
    } else {
        for (int n = lane; n < n2; n += 64) T->g_v[n] = P->g_v[n];
        if (lane < k) T->c_v[lane] = P->c_v[lane];
    }
    // what a worker asks before it starts (or goes on with) a task of level-3 node t - ONE byte instead of three dependent looks
    for (int t = lane; t < total; t += 64) {
        const int te = P->task_entry[t];
        const bool dead = P->result[t] != RAZ_SOLVER_UNKNOWN || (!exact && (P->g_v[te & 0xff] != RAZ_SOLVER_UNKNOWN || P->c_v[te >> 8] != RAZ_SOLVER_UNKNOWN));
        if (dead) D->h_dead[t] = 1;
    }
    // A win/loss solve (the reference's `not exactly`: a node's loop ends at the first move with a value > 0) needs the root's moves
    // one after the other: the subtrees below the third open move matter only if the first two both fail, and the pool spends 21 500
    // lane-iterations per solve where the reference's own search visits ~1 250 nodes.  RAZ_SOLVER_NE_WINDOW = n releases only the tasks
    // below the first n open root moves and moves the window on as scans decide them.  MEASURED, round 6 (mini.yml as shipped, one box,
    // profiles/r6/solver_win_loss_task_window_ab.jsonl): 0 / 1 / 2 / 3 open moves = 21.8 / 13.2 / 17.3 / 19.4 M sims/s lock-step, 27.5 /
    // 18.1 / 21.9 / 24.3 M with continuous batching - the busy lane-iterations fall by 6 % only (the discarded work sits below the
    // replies, not below the root's later moves) while every solve needs more rounds.  Kept as a compile-time knob, off.
    // (the limit is a LEVEL-3 NODE number, read off the tree in LDS: tasks of the nodes [0, limit) are released)
    uint32_t limit = (uint32_t)total;
    if (!exact && RAZ_SOLVER_NE_WINDOW > 0) {
        int open = 0, last = k - 1;
        for (int i = 0; i < k; ++i)
            if (P->c_v[i] == RAZ_SOLVER_UNKNOWN && ++open == RAZ_SOLVER_NE_WINDOW) {
                last = i;
                break;
            }
        limit = (uint32_t)uni((uint32_t)P->g_first[P->c_first[last + 1]]);
    }
    if (lane == 0) h->limit = limit;
    if (collect) {
        const uint32_t next = st == RAZ_SOLVE_REQUESTED ? 0u : uni(h->next);
        if (next < (uint32_t)subs && lane == 0) E.pool_active[g0 + atomicAdd(&ph->n_active, 1u)] = g;
        if (lane == 0) {
            h->rounds = (st == RAZ_SOLVE_REQUESTED ? 0u : h->rounds) + 1u;
            h->rounds_total += 1u;
            if (st == RAZ_SOLVE_REQUESTED) atomicAdd(&E.counters[20], 1ULL);
        }
    }
}

// ------------------------------------------------------------------ k_solve_run
// Worker wave w of the pool: lane state in E.pool_state[w][word][lane] - 0 own, 1 enemy, 2 moves left, 3 the search's small fields,
// 4 game slot | request generation << 32, 5 task | level-2 node << 16 | child << 24 - and frames in E.pool_frames[w][level][lane].
// The lanes consult the memo only at nodes with at least RAZ_SOLVER_LANE_MEMO_EMPTIES empties (with the fourth ply of tasks a 10-empties
// solve's subtrees start at 6: its workers do not touch the memo at all, its scans do): the 64 searches advance in lockstep,
// so ONE lane's probe (dependent HBM round trips into a 2 MB table) is paid by all of them; subtrees below that size are searched
// outright (<= 720 leaf paths) - and even those probes are batched (the slow phase, below).
#ifndef RAZ_SOLVER_LANE_MEMO_EMPTIES
#define RAZ_SOLVER_LANE_MEMO_EMPTIES 7   // (6: 23.3 M sims/s on mini.yml as shipped, 7: 24.5 M - profiles/r5/solver_worker_loop_compile_time_knobs_ab.jsonl)
#endif
#ifndef RAZ_SOLVER_MAX_RETURNS
#define RAZ_SOLVER_MAX_RETURNS 2
#endif
#ifndef RAZ_SOLVER_PROBE_AT_DRAW
#define RAZ_SOLVER_PROBE_AT_DRAW 0
#endif
#ifndef RAZ_SOLVER_SLOW_EVERY
#define RAZ_SOLVER_SLOW_EVERY 16   // (a power of two; 4 / 8 / 16: 22.6 / 23.3 / 23.1 M at memo 6, 24.8 M at 16 with memo 7)
#endif
#ifndef RAZ_SOLVER_POOL_EVERY
#define RAZ_SOLVER_POOL_EVERY 1   // tree launches per round of the pool (raz_engine.hip: > 1 = the round runs beside them on its own stream)
#endif
#ifndef RAZ_SOLVER_POOL_BUDGET
#define RAZ_SOLVER_POOL_BUDGET 96   // (128 until the move function finished the last square itself: see RAZ_SOLVER_INLINE_LAST)
#endif
#ifdef RAZ_WAVE_EMU
#define RAZ_POOL_WAVES
#else
#define RAZ_POOL_WAVES __attribute__((amdgpu_waves_per_eu(4, 4)))
#endif
__global__ __launch_bounds__(64) RAZ_POOL_WAVES void k_solve_run(raz_engine_dev E, uint32_t part, uint32_t w0, uint32_t wcount, uint32_t a0, int budget) {
    if (blockIdx.x >= wcount) return;
    const int lane = threadIdx.x;
    const bool conc = E.xk != 0u;   // the tree kernels may be running beside this launch (raz_engine_core.h xk_*)
    const uint32_t w = w0 + blockIdx.x;
    unsigned long long* lw = E.pool_state + (size_t)w * 1024;                  // lw[word * 64 + lane]
    // the lanes' frames live in LDS while the wave runs (a push or a pop is four conflict-free 8-byte accesses - word j of level d at
    // [(d * 4 + j) * 64 + lane] - instead of an HBM round trip in the middle of every node) and in E.pool_frames between launches
    __shared__ unsigned long long frames_lds[RAZ_SOLVER_MAX_DEPTH * 4 * 64];
    unsigned long long* fr = frames_lds + lane;                                                            // word j of level d: fr[(d * 4 + j) * 64]
    unsigned long long* fr_hbm = E.pool_frames + (size_t)w * RAZ_SOLVER_MAX_DEPTH * 4 * 64 + (size_t)lane;   // the same layout
    raz_solver_pool_hdr* ph = E.pool_hdr + part;
    const uint32_t nact = uni(ph->n_active);
    const uint32_t* active = E.pool_active + a0;
    const unsigned long long m1 = lw[3 * 64 + lane];
    bool have = (m1 & 1ULL) != 0;
    if (__ballot(have) == 0ULL && nact == 0u) return;   // nothing parked here, nothing to hand out
    raz_bb own = lw[0 * 64 + lane], enemy = lw[1 * 64 + lane], left = lw[2 * 64 + lane];
    int fresh = (int)((m1 >> 1) & 1ULL), flip = (int)((m1 >> 2) & 1ULL), task_sign = ((m1 >> 3) & 1ULL) ? -1 : 1;
    uint32_t exact = (uint32_t)((m1 >> 4) & 1ULL);
    int d = (int)((m1 >> 8) & 0xffULL), bmv = (int)((m1 >> 16) & 0xffULL) - 1, bsc = (int)((m1 >> 24) & 0xffULL) - 128, pact = (int)((m1 >> 32) & 0xffULL) - 1;
    uint32_t g = (uint32_t)lw[4 * 64 + lane], gen = (uint32_t)(lw[4 * 64 + lane] >> 32);
    const unsigned long long m2 = lw[5 * 64 + lane];
    int task = (int)(m2 & 0xffffULL), task_n = (int)((m2 >> 16) & 0xffULL), task_ci = (int)((m2 >> 24) & 0xffULL), task_t = (int)((m2 >> 32) & 0xffffULL);
    if (have) {   // is the parked search still wanted?  Its request may have been answered (a decided scan) or replaced, its node decided
        const raz_solve_hdr* hh = solve_hdr(E, g);
        const unsigned long long sg = xk_load64(conc, (const unsigned long long*)hh);   // {state, gen}: one word, as the tree kernels publish it
        if ((uint32_t)(sg >> 32) != gen || RAZ_SOLVE_STATE((uint32_t)sg) != RAZ_SOLVE_RUNNING) have = false;
        else if (solve_deep(E, g)->h_dead[task_t]) have = false;
    }
    if (have)   // (a lane reads and writes its own column only: no barrier)
        for (int i = 0; i < d * 4; ++i) fr[i * 64] = fr_hbm[i * 64];
    bool dry = nact == 0u;
    int empty_draws = 0;
    uint32_t st_busy = 0u, st_iters = 0u, st_done = 0u, st_skipped = 0u;   // statistics (raz_engine_solver_stats)
    unsigned long long tk_slow = 0, tk_pop = 0;
    const unsigned long long tk_begin = prof_now();
    // A lane's memo traffic waits for the wave's next SLOW PHASE (every RAZ_SOLVER_SLOW_EVERY-th iteration): the 64 searches run in
    // lockstep, so a probe made the moment one lane reaches a node of >= 6 empties - some lane does in nearly every iteration -
    // stalled ALL of them for an HBM round trip per iteration (first hardware runs: 5 us per iteration, 0.75 us of it arithmetic).
    // In the slow phase the wanted probes and the claims of finished nodes are issued first, the idle lanes' task draws (five
    // dependent round trips) run on top of them, and the probes' answers are looked at last.  A lane that wants a probe idles until
    // then (a few iterations at a node that has ~1 chance in 30 of being such a node).
    bool wait_find = ((m1 >> 5) & 1ULL) != 0;
    const uint32_t lane_id = blockIdx.x * 64u + (uint32_t)lane, lanes_of_slice = wcount * 64u;
    uint32_t draws = 0u, next_solve = nact ? active[lane_id % nact] : 0u;
    // how many tasks the solve of the lane's next draw had left when last looked at (requested a phase ahead, like the list entry): a lane
    // only draws - an atomic on the solve's counter - when its turn among the lanes that walk to that solve comes up often enough for the
    // tasks that are left.  Without this every idle lane of the slice drew at every slow phase: with a handful of solves listed (the
    // first and the last few hundred steps of a batch) that is 65 000 atomics on a handful of addresses, ~88 per microsecond each -
    // rounds of 1 ms that hand out a few hundred tasks (tools/solver_timeline.py)
    const uint32_t lanes_per_solve = nact ? (lanes_of_slice + nact - 1u) / nact : 1u;
    uint32_t peek_left = 0u;
    if (nact) {
        const raz_solve_hdr* hp = solve_hdr(E, next_solve);
        const uint32_t pn = __hip_atomic_load(&hp->next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), pt = hp->total;
        peek_left = pn < pt ? pt - pn : 0u;
    }
    bool put_pending = false;
    raz_bb put_own = 0, put_enemy = 0;
    uint32_t put_tag = 0u, put_g = 0u;
    for (int iter = 0; iter < budget; ++iter) {
        const unsigned long long idle = __ballot(!have);
        const int nidle = __popcll(idle);
        st_busy += 64u - (uint32_t)nidle;
        ++st_iters;
        bool hit = false;
        int hit_m = 0, hit_s = 0;
        const unsigned long long tk0 = prof_now();
        if ((iter & (RAZ_SOLVER_SLOW_EVERY - 1)) == 0) {
            // ---- 1. memo traffic goes out
            const bool finding = have && wait_find;
            uint32_t it0 = 0u, it1 = 0u, claimed = 1u;
            raz_bb b0 = 0, w0 = 0, b1 = 0, w1 = 0;
            raz_slot* pslot = nullptr;
            if (finding) {
                const raz_slot* tab = E.memo + (size_t)g * E.M;
                const uint32_t h = key_hash(own, enemy, 8u + exact);
                const raz_slot *s0 = tab + (h & (E.M - 1)), *s1 = tab + ((h + 1u) & (E.M - 1));
                it0 = xk_load32(conc, &s0->idx_tag); it1 = xk_load32(conc, &s1->idx_tag);   // (memo entries cross between concurrent kernels: raz_engine_core.h xk_*)
                b0 = xk_load64(conc, &s0->black); w0 = xk_load64(conc, &s0->white); b1 = xk_load64(conc, &s1->black); w1 = xk_load64(conc, &s1->white);
            }
            if (put_pending) {
                pslot = E.memo + (size_t)put_g * E.M + (key_hash(put_own, put_enemy, 8u + ((put_tag >> 30) & 1u)) & (E.M - 1));
                claimed = atomicCAS(&pslot->idx_tag, 0u, RAZ_MEMO_CLAIMED);
            }
            // ---- 2. idle lanes draw tasks.  No shared cursor (one address takes ~88 atomics per microsecond on this chip: a cursor drawn
            // by every wave every few iterations was the pool's ceiling): lane L of wave w walks its own sequence over the active list,
            // solve (w * 64 + L + q * lanes of the slice) mod nact at its q-th draw - all lanes together sweep the list evenly, and a solve's
            // tasks still leave in scan order (its `next` counter).  The list entry of a lane's next draw is requested one phase ahead
            if (!dry && nidle) {
                bool got = false, saw_tasks = false;
                // (RAZ_SOLVER_DRAW_GATE: of the lanes_per_solve lanes that walk to this solve, about as many draw as it has tasks left -
                // a lane's turn comes round with its draws)
                const uint32_t every = peek_left ? (lanes_per_solve + peek_left - 1u) / peek_left : 0u;
                const bool my_turn = !RAZ_SOLVER_DRAW_GATE || (every && (lane_id / nact + draws) % every == 0u);
                if (!have) {
                    const uint32_t gg = next_solve;
                    saw_tasks = peek_left != 0u;
                    ++draws;
                    next_solve = active[(lane_id + draws * lanes_of_slice) % nact];
                    {
                        const raz_solve_hdr* hp = solve_hdr(E, next_solve);
                        const uint32_t pn = __hip_atomic_load(&hp->next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), pt = hp->total;
                        peek_left = pn < pt ? pt - pn : 0u;
                    }
                    if (my_turn) {
                    raz_solve_hdr* hh = solve_hdr(E, gg);
                    const uint32_t t = atomicAdd(&hh->next, 1u);
                    const uint32_t total = hh->total, window = hh->limit, ex = xk_load32(conc, &hh->exact), hgen = xk_load32(conc, &hh->gen);   // (fixed while the pool runs: requested beside the draw; exact and gen are the tree kernels' words)
                    const int nt_ = t < total ? (int)solve_deep(E, gg)->sub_task[t] : 0;
                    const bool beyond = RAZ_SOLVER_NE_WINDOW > 0 && t < total && (uint32_t)nt_ >= window;
                    // A ticket outside this round's window (k_solve_scan's limit) goes back for the round that opens it - and so does one past
                    // the end of the list: thousands of lanes draw at once, and a ticket that stayed drawn behind the window's end would be a
                    // task nobody ever searches (first hardware run of the window: the batch did not finish; the emulator's waves draw one
                    // after the other and never overshoot that far).  Tickets inside the window are handed out once each: the counter only
                    // ever comes back down to the window's end (every subtraction undoes an addition that got a ticket >= that end).
                    if (RAZ_SOLVER_NE_WINDOW > 0 && (beyond || t >= total)) atomicSub(&hh->next, 1u);   // (without a window a ticket past the end just stays drawn)
                    if (t < total && !beyond) {
                        got = true;
                        SolverTree* T = solve_tree(E, gg);
                        SolverDeep* D = solve_deep(E, gg);
                        const int nt = nt_;   // the task's level-3 node
                        const int te = T->task_entry[nt], n = te & 0xff, ci = te >> 8, hk = D->h_kind[nt], s0 = D->sub_first[nt];
                        const raz_bb ho = D->h_own[nt], he = D->h_enemy[nt], hm = D->h_moves[nt];
                        // (a task that has its result: dispatch restarted after the pool was re-partitioned; a node that is decided: moot)
                        if (D->sub_result[t] == RAZ_SOLVER_UNKNOWN && !D->h_dead[nt]) {
                            raz_bb no = ho, ne = he, nm = hm;
                            int v = 0, kind = 2;   // an unsplit node: the search starts at the node itself (same mover: sign +)
                            if (!(hk & 8)) {
                                raz_bb m = hm;
                                for (int j = s0; j < (int)t; ++j) m &= m - 1;
                                kind = solver_play(__ffsll((long long)m) - 1, ho, he, no, ne, nm, v, ex != 0u);
                            }
                            int pm = 0, ps = 0;
                            if (RAZ_SOLVER_PROBE_AT_DRAW && kind && bb_popcount(~(no | ne)) >= RAZ_SOLVER_LANE_MEMO_EMPTIES && memo_find_lane(E, gg, no, ne, ex, pm, ps)) {
                                // the memo knows the subtree's root (a transposition another lane searched): the probe a search starts with,
                                // made here, where the wave is waiting on memory anyway
                                D->sub_result[t] = (signed char)((kind == 1 ? -1 : 1) * ps);
                                ++st_done;
                            } else if (kind) {
                                g = gg;
                                gen = hgen;
                                exact = ex;
                                task = (int)t;
                                task_t = nt;
                                task_n = n;
                                task_ci = ci;
                                task_sign = kind == 1 ? -1 : 1;
                                d = 0;
                                own = no;
                                enemy = ne;
                                left = nm;
                                bmv = -1;
                                bsc = -100;
                                pact = -1;
                                flip = 0;
                                fresh = RAZ_SOLVER_PROBE_AT_DRAW ? 0 : 1;   // (its root was looked up in the memo just now)
                                wait_find = false;
                                have = true;
                            } else {
                                D->sub_result[t] = (signed char)v;
                                ++st_done;
                            }
                        } else
                            ++st_skipped;
                    }
                    }
                }
                if (__ballot(got) == 0ULL) {
                    // (a phase in which lanes saw tasks but it was nobody's turn does not count: the tasks are still there)
                    if (!(RAZ_SOLVER_DRAW_GATE && __ballot(!got && saw_tasks && !my_turn)) && ++empty_draws >= 2) dry = true;   // the listed solves have handed out everything (this launch)
                } else
                    empty_draws = 0;
            }
            // ---- 3. the memo's answers
            if (finding) {
                wait_find = false;
                fresh = 0;   // (probed: a miss goes on with the node's moves)
                uint32_t it = 0u;
                if ((it0 >> 31) && b0 == own && w0 == enemy && ((it0 >> 30) & 1u) == exact) it = it0;
                else if ((it0 >> 31) && (it1 >> 31) && b1 == own && w1 == enemy && ((it1 >> 30) & 1u) == exact) it = it1;
                if (it) {
                    hit = true;
                    hit_m = (int)((it >> 8) & 0xffu) - 1;
                    hit_s = (int)(it & 0xffu) - 128;
                }
            }
            // (the home slot only: a finished node whose slot is taken is simply not remembered.)  memo_claim_and_write's second half:
            // keys, drained, then the tag - all agent-scope words; the wait is the wave's, not a lane's, so it stands outside the branch
            const bool putting = put_pending && claimed == 0u;
            if (putting) {
                xk_store64(conc, &pslot->black, put_own);
                xk_store64(conc, &pslot->white, put_enemy);
            }
            if (__ballot(putting)) xk_drain(conc);   // (only a wave that stores keys waits: the wait also covers the loads requested a phase ahead)
            if (!conc) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // (compiler order)
            if (putting) __hip_atomic_store(&pslot->idx_tag, put_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            put_pending = false;
        }
        const unsigned long long tk1 = prof_now();
        tk_slow += tk1 - tk0;
        if (__ballot(have) == 0ULL) {
            if (dry) break;
            continue;
        }
        if (have && !wait_find) {
            // ---- one MOVE of this lane's search per iteration (solver_solve_scalar's loop, its "node is finished" steps folded in):
            // first every node that is finished hands its value to its parent - cheap LDS pops, usually none or one - then the node the
            // search stands on plays its next move.  (A separate iteration per finished node made 45 % of the iterations pops, each at
            // the full price of the lockstep wave's expand path.)
            // At most RAZ_SOLVER_MAX_RETURNS returns per iteration: the wave runs as many rounds of this loop as its DEEPEST unwinding
            // lane needs (unbounded: 4-5 rounds, 28 % of the wave's time by the tick counters); a lane that has more to unwind goes on
            // at the next iteration without playing a move in this one (about one lane-iteration in 25).
            bool may_move = false;
            for (int returns = 0; returns <= RAZ_SOLVER_MAX_RETURNS; ++returns) {
                const bool big = bb_popcount(~(own | enemy)) >= RAZ_SOLVER_LANE_MEMO_EMPTIES;
                int rm = 0, rs = 0;
                bool done = false;
                if (hit) {
                    rm = hit_m;
                    rs = hit_s;
                    done = true;
                    hit = false;
                    fresh = 0;
                } else if (fresh && big) {
                    wait_find = true;   // the probe goes out with the next slow phase
                    break;
                } else {
                    fresh = 0;
                    if (left == 0 || (!exact && bsc > 0)) {
                        if (big) {   // remembered at the next slow phase (one finished node per lane at a time)
                            put_pending = true;
                            put_own = own;
                            put_enemy = enemy;
                            put_g = g;
                            put_tag = 0x80000000u | (exact << 30) | ((uint32_t)(bmv + 1) << 8) | (uint32_t)(bsc + 128);
                        }
                        rm = bmv;
                        rs = bsc;
                        done = true;
                    }
                }
                if (!done) {
                    may_move = true;
                    break;
                }
                if (returns == RAZ_SOLVER_MAX_RETURNS) break;   // (finished, but its return waits for the next iteration)
                if (d == 0) {
                    solve_deep(E, g)->sub_result[task] = (signed char)(task_sign * rs);
                    have = false;
                    ++st_done;
                    break;
                }
                // back to the parent
                const int v = flip ? -rs : rs, a = pact;
                --d;
                own = fr[(d * 4 + 0) * 64];
                enemy = fr[(d * 4 + 1) * 64];
                left = fr[(d * 4 + 2) * 64];
                const uint32_t meta = (uint32_t)fr[(d * 4 + 3) * 64];
                bmv = (int)(meta & 0xffu) - 1;
                bsc = (int)((meta >> 8) & 0xffu) - 128;
                pact = (int)((meta >> 16) & 0xffu) - 1;
                flip = (int)((meta >> 24) & 1u);
                if (bsc < v) {
                    bmv = a;
                    bsc = v;
                }
            }
            tk_pop += prof_now() - tk1;
            if (have && !wait_find && may_move) {
                const int a = __ffsll((long long)left) - 1;
                left &= left - 1;
                raz_bb no, ne, nm;
                int score;
                const int kind = solver_play(a, own, enemy, no, ne, nm, score, exact != 0u);
                if (kind) {   // down a ply
                    fr[(d * 4 + 0) * 64] = own;
                    fr[(d * 4 + 1) * 64] = enemy;
                    fr[(d * 4 + 2) * 64] = left;
                    fr[(d * 4 + 3) * 64] = (unsigned long long)((uint32_t)(bmv + 1) | ((uint32_t)(bsc + 128) << 8) | ((uint32_t)(pact + 1) << 16) | ((uint32_t)flip << 24));
                    ++d;
                    own = no;
                    enemy = ne;
                    left = nm;
                    bmv = -1;
                    bsc = -100;
                    pact = a;
                    flip = kind == 1 ? 1 : 0;
                    fresh = 1;
                } else if (bsc < score) {
                    bmv = a;
                    bsc = score;
                }
            }
        }
    }
    {
        const uint32_t done = wave_sum_u32(st_done), skipped = wave_sum_u32(st_skipped);
        if (lane == 0) {
            atomicAdd(&E.counters[23], (unsigned long long)st_busy);
            atomicAdd(&E.counters[24], (unsigned long long)st_iters);
            atomicAdd(&E.counters[25], (unsigned long long)done);
            atomicAdd(&E.counters[26], (unsigned long long)skipped);
            atomicAdd(&E.counters[27], 1ULL);
            atomicAdd(&E.counters[17], tk_slow);                    // shader-clock ticks: slow phases,
            atomicAdd(&E.counters[18], tk_pop);                     // the pops of finished nodes,
            atomicAdd(&E.counters[19], prof_now() - tk_begin);      // the whole loop
        }
    }
    // park: the next launch goes on from here
    if (have)
        for (int i = 0; i < d * 4; ++i) fr_hbm[i * 64] = fr[i * 64];
    lw[0 * 64 + lane] = own;
    lw[1 * 64 + lane] = enemy;
    lw[2 * 64 + lane] = left;
    lw[3 * 64 + lane] = (have ? 1ULL : 0ULL) | ((unsigned long long)(fresh & 1) << 1) | ((unsigned long long)(flip & 1) << 2) | ((task_sign < 0 ? 1ULL : 0ULL) << 3) |
                        ((unsigned long long)(exact & 1u) << 4) | ((wait_find ? 1ULL : 0ULL) << 5) | ((unsigned long long)(d & 0xff) << 8) | ((unsigned long long)((bmv + 1) & 0xff) << 16) |
                        ((unsigned long long)((bsc + 128) & 0xff) << 24) | ((unsigned long long)((pact + 1) & 0xff) << 32);
    lw[4 * 64 + lane] = (unsigned long long)g | ((unsigned long long)gen << 32);
    lw[5 * 64 + lane] = (unsigned long long)(task & 0xffff) | ((unsigned long long)(task_n & 0xff) << 16) | ((unsigned long long)(task_ci & 0xff) << 24) |
                        ((unsigned long long)(task_t & 0xffff) << 32);
}

// per-game totals for raz_engine_solver_stats: counters[28] solves of all games, [29] most solves of one game, [30] rounds all games'
// solves were listed in, [31] most of one game (its critical path in rounds of the pool)
__global__ __launch_bounds__(256) void k_solver_game_stats(raz_engine_dev E) {
    __shared__ unsigned long long sh[4][256];
    unsigned long long a = 0, b = 0, c = 0, d = 0;
    for (uint32_t g = threadIdx.x; g < E.B; g += 256) {
        const raz_solve_hdr* h = solve_hdr(E, g);
        const unsigned long long n = h->posted, r = h->rounds_total;   // (not gen: k_solve_pool_reset bumps that too)
        a += n;
        b = n > b ? n : b;
        c += r;
        d = r > d ? r : d;
    }
    sh[0][threadIdx.x] = a; sh[1][threadIdx.x] = b; sh[2][threadIdx.x] = c; sh[3][threadIdx.x] = d;
    __syncthreads();
    if (threadIdx.x < 4) {
        unsigned long long x = 0;
        for (int i = 0; i < 256; ++i) x = (threadIdx.x & 1) ? (sh[threadIdx.x][i] > x ? sh[threadIdx.x][i] : x) : x + sh[threadIdx.x][i];
        E.counters[28 + threadIdx.x] = x;
    }
}

// after the batch was re-partitioned into a different number of slices (raz_engine_set_parts): parked searches may sit in another
// slice's part of the pool than their game - drop them all and hand every running solve's tasks out again (the ones that have their
// result are skipped at the draw).  One thread per worker lane / per game.
__global__ __launch_bounds__(256) void k_solve_pool_reset(raz_engine_dev E) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < (size_t)E.W * 64) {
        E.pool_state[(i / 64) * 1024 + 3 * 64 + (i % 64)] = 0ULL;
        E.pool_state[(i / 64) * 1024 + 9 * 64 + (i % 64)] = 0ULL;
    }
    if (i < E.B) {
        raz_solve_hdr* h = solve_hdr(E, (uint32_t)i);
        if (RAZ_SOLVE_STATE(h->state) == RAZ_SOLVE_RUNNING) {
            h->gen += 1u;
            h->next = 0u;
        }
    }
}

}  // namespace
