// raz_engine_fused.hip — tree + net in ONE kernel for narrow nets: k_tree_net.
//
// With the F == 16 net (mini.yml) a step of the classic pipeline is two latency-bound launches per slice - k_tree (25 us) and
// k_net_mfma (one wave per position, 13-25 us) - handing leaves and answers over through HBM, and its rate is set by that chain's
// latency and by how many launches per second one host thread and the hardware queues sustain (more than three
// slices or hipGraph replay made it slower).  Here the game's wave evaluates its own leaf: the descent ends with the position
// in registers, raz_net16_forward_in_wave (raz_net_wave.h: matrix-core trunk in one LDS plane buffer of this wave, heads as
// k_net_mfma) returns the policy row and the value into the registers backup_leaf reads, and the loop goes on with the next
// simulation - `iters` of them per launch, the control block and the path staying in registers throughout.  No leaf exchange, no
// second kernel, no slices; one launch per `iters` simulation steps of the whole batch.  4 waves per SIMD (128 VGPRs, 9.7 KB of
// LDS per wave): 4096 games are resident at once.  Measured resources (-Rpass-analysis=kernel-resource-usage): k_tree_net<false>
// spills 31 VGPRs (128 B of scratch per lane) around the in-wave net call - most of the kernel's counter traffic
// (profiles/r4_pmc/config1_fused_*), and cheaper than not spilling (below) - k_tree_par_net<false> none; the SOLVER forms are at 4 waves per SIMD too since round 5 (the end-game search
// itself runs in the solver pool, raz_solver_pool.h: these kernels only post positions and solve <= 4 empties in place; a game whose
// request is in flight leaves the launch, and the pool gets its round after every <= 8 fused steps).  Every game performs exactly the
// operations it performs under k_tree + k_net_mfma, in the same order: results are bit-identical (tests/test_engine_fused_emu.py
// on the wave emulator, tests/test_engine_gpu.py, tests/test_zz_fused_gpu.py).  raz_engine_config.reserved bit 4, without the
// evaluation cache; the worker's default for 16-filter nets when the solver is off (105 M against 77 M sims/s on BASELINE
// configs[1]).  At the end of a launch the last leaf's answer is parked in nn_policy / nn_value exactly where k_net_mfma would
// have put it, so the two forms can even alternate.
#include <hip/hip_runtime.h>
#include "raz_engine_core.h"
#include "raz_net_wave.h"   // raz_net16_forward_in_wave

#ifndef RAZ_FUSED_ITERS
#define RAZ_FUSED_ITERS 256   // most simulation steps per launch (32 -> 256: +0.9 % on configs[1], profiles/r5/fused_kernel_ab.json)
#endif

namespace {

// Loop-invariant values that the compiler computes once before the step loop and keeps in VGPRs for the whole launch - the lane's
// addresses and masks, (double)virtual_loss, 1 / dirichlet_alpha, (float)c_puct ... - are the registers it spills around the in-wave
// net call (128 VGPRs per wave; the forward needs most of them).  Taking the lane id and those config words through an empty asm
// once per step (or per round operation) makes it recompute them where they are used instead, and the spills go away: measured on
// one box, configs[1] whole games (tools/sessions/r5_s17.sh, profiles/r5/fused_kernel_ab.json):
//     k_tree_par_net<false>  56 spilled VGPRs -> 0    92.3 -> 96.3 M sims/s   (kept: RAZ_FRESH_K = 7)
//     k_tree_net<false>      30 spilled VGPRs -> 0   105.3 -> 101.3 M sims/s  (NOT kept: RAZ_FRESH_1 = 0.  The kernel is bound by
//         instruction issue, 4 waves x 21 % per SIMD; its spills are one batch of scratch stores before the forward and one batch of
//         loads behind it, which cost less than the recomputation - lane id alone -2.5 %, config words alone -0.6 %)
// bit 0: lane id per step, bit 1: config words per step, bit 2 (k_tree_par_net): lane id per round operation and per queued leaf.
// bit 3 (round 6): the engine's descriptor read from the kernel-argument segment where it is used (fresh_descriptor, raz_engine_core.h)
// instead of the config words' asm (bit 1 is then without effect): k_tree_net<false> 206 -> 121 spilled scalar registers, 31 -> 21
// spilled vector registers, 3985 -> 3670 vector instructions in the kernel; k_tree_par_net<false> 256 -> 193, 4547 -> 4020.
// configs[1] whole games, two runs each on one box (tools/sessions/r6_s15.sh, profiles/r6/fused_kernels_descriptor_from_the_kernarg_segment_ab.json):
//     k_tree_net<false>      105.3-105.8 -> 107.0-107.4 M sims/s;  k_tree_par_net<false>  96.8 -> 98.7-99.0 M;  mini.yml as shipped 21.6 -> 22.0-22.2 M
#ifndef RAZ_FRESH_1
#define RAZ_FRESH_1 8
#endif
#ifndef RAZ_FRESH_K
#define RAZ_FRESH_K 15
#endif
// -DRAZ_FUSED_PROF (measurement builds only): shader-clock ticks of a step's phases of k_tree_net into the engine's phase profile
// (engine.phase_profile(): 0 backup, 1 controller, 2 select, 5 the in-wave forward; the engine created with phase_profile=True)
#ifdef RAZ_FUSED_PROF
#define RAZ_FUSED_T0() unsigned long long t_prof = prof_now()
#define RAZ_FUSED_T(k) do { prof_add(E, g, (k), t_prof, lane); t_prof = prof_now(); } while (0)
#else
#define RAZ_FUSED_T0() ((void)0)
#define RAZ_FUSED_T(k) ((void)0)
#endif
template <int ON = 1>
__device__ __forceinline__ int fresh_lane(int lane) {
#ifndef RAZ_WAVE_EMU
    if (ON) asm volatile("" : "+v"(lane));
#endif
    return lane;
}
template <int ON = 1>
__device__ __forceinline__ raz_engine_dev fresh_config(const raz_engine_dev& E) {
    raz_engine_dev F = E;
#ifndef RAZ_WAVE_EMU
    if (ON) asm volatile("" : "+s"(F.cfg.virtual_loss), "+s"(F.cfg.required_visit_to_decide_action), "+s"(F.cfg.c_puct), "+s"(F.cfg.noise_eps), "+s"(F.cfg.dirichlet_alpha));
#endif
    return F;
}

template <bool SOLVER>
__global__ __launch_bounds__(64) RAZ_TREE_WAVES(SOLVER) void k_tree_net(raz_engine_dev E, uint32_t g0, uint32_t count,
                                                                                          uint32_t iters, const float* __restrict__ net_w,
                                                                                          int net_R, int net_V) {
    if (blockIdx.x >= count) return;
    // LDS of the wave: the net's plane buffer, then ONE scratch area shared in time by the net's heads (during a forward), the
    // reduction scratch of backup_leaf and the solver's frames (between forwards; both are initialised by their users on every
    // call) - 9.7 KB in all, so that 16 waves fit a CU's 160 KB whatever the allocation granule
    extern __shared__ __attribute__((aligned(16))) float netbuf[];
    float* lds64 = netbuf + 16 * PS;
    SolverLDS* slds_p = SOLVER ? (SolverLDS*)(netbuf + 16 * PS + 64) : nullptr;
    const uint32_t g = g0 + blockIdx.x;
    int lane = threadIdx.x;
    if (g >= E.B) return;
    if (SOLVER && solve_in_flight(E, g)) return;   // the request is still with the solver pool: nothing to do in this launch
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
        if (phase == RAZ_PHASE_DONE || phase == RAZ_PHASE_IDLE || G32(R, GW(error))) return;
    }
    raz_net16_zero_planes(netbuf, lane);
    const int lane0 = lane;
    const raz_engine_dev& E0 = E;
    for (uint32_t it = 0; it < iters; ++it) {
        const int lane = fresh_lane<(RAZ_FRESH_1 & 1)>(lane0);
#if (RAZ_FRESH_1 >> 3) & 1
        const raz_engine_dev& E = fresh_descriptor<1>(E0);
#else
        const raz_engine_dev E = fresh_config<(RAZ_FRESH_1 >> 1) & 1>(E0);
#endif
        uint32_t phase = G32(R, GW(phase));
        if (phase == RAZ_PHASE_DONE || phase == RAZ_PHASE_IDLE) break;
        if (G32(R, GW(error))) break;
        // a descent suspended at an in-simulation solve (select_leaf) goes on where it stands, before anything else
        const bool suspended = SOLVER && G32(R, GW(leaf_kind)) == RAZ_LEAF_SOLVE_PENDING;
        RAZ_FUSED_T0();
        if (!suspended) {
            if (G32(R, GW(leaf_kind)) != RAZ_LEAF_NONE) backup_leaf<false>(E, R, g, G32(R, GW(player)) - 1, lane, lds64);
            RAZ_FUSED_T(0);
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
            RAZ_FUSED_T(1);
            phase = G32(R, GW(phase));
            if (phase != RAZ_PHASE_SEARCH || (int32_t)G32(R, GW(sims_left)) <= 0 || G32(R, GW(error))) break;
        }
        select_leaf<SOLVER, false>(E, R, g, lane, slds_p, g, suspended ? G32(R, GW(leaf_node)) : G32(R, GW(root_node)),
                                   suspended ? (int)G32(R, GW(depth)) : 0, false, suspended ? (int)G32(R, GW(leaf_action)) - 1 : -1);
        RAZ_FUSED_T(2);
        const uint32_t lk = G32(R, GW(leaf_kind));
        if (lk == RAZ_LEAF_EXPAND) {
            // what select_leaf handed to the leaf exchange (nn_own / nn_enemy), recomputed from the control block: the
            // leaf's position under the D4 transform drawn for it, from the side to move's view (player.py:299-309)
            const uint32_t sym = G32(R, GW(leaf_sym));
            const raz_bb lb = G64(R, GW(leaf_b)), lw = G64(R, GW(leaf_w));
            const raz_bb tb = bb_d4_apply(lb, (int)(sym >> 2) & 1, (int)(sym & 3)), tw = bb_d4_apply(lw, (int)(sym >> 2) & 1, (int)(sym & 3));
            const bool black_to_move = G32(R, GW(leaf_np)) == 1u;
            raz_net16_forward_in_wave(net_w, net_R, net_V, black_to_move ? tb : tw, black_to_move ? tw : tb, netbuf, lane, R.pol_raw, R.val);
            R.nn = 0u;
            RAZ_FUSED_T(5);
        } else if (lk != RAZ_LEAF_TERMINAL && lk != RAZ_LEAF_SOLVED)
            break;
    }
    lane = fresh_lane<(RAZ_FRESH_1 & 1)>(lane);
    gw[lane] = R.cw;
    if (R.path_dirty) path_store(E, R, (size_t)g, lane);
    // the answer for a leaf that is still to be backed up waits where the net kernel would have left it
    E.nn_policy[(size_t)g * 64 + lane] = R.pol_raw;
    if (lane == 0) {
        E.nn_value[g] = R.val;
        E.nn_active[g] = 0;
    }
}

// The same for parallel_search_num > 1: k_tree_par's round (B: the queued leaves' simulations return, C: the free slots are
// refilled, D: sleepers are polled, C': refill) followed by the evaluation of the round's queued leaves by the game's own wave, one
// after the other, `iters` times per launch.  One iteration is exactly one launch of k_tree_par + the net batch of the classic
// pipeline - including a fill that ran out of its per-launch budget and goes on after the evaluation without a B in between - so
// the raz-sched-v1 schedule, and with it every record, is unchanged.  The slot states stay in registers across iterations; the
// leaves' positions and answers still travel through the leaf-exchange rows (written and read by this wave only).
template <bool SOLVER>
__global__ __launch_bounds__(64) RAZ_TREE_WAVES(SOLVER) void k_tree_par_net(raz_engine_dev E, uint32_t g0, uint32_t count,
                                                                                              uint32_t iters, const float* __restrict__ net_w,
                                                                                              int net_R, int net_V) {
    if (blockIdx.x >= count) return;
    // LDS of the wave: the net's plane buffer, then ONE scratch area shared in time by the net's heads (during a forward), the
    // reduction scratch of backup_leaf and the solver's frames (between forwards; both are initialised by their users on every
    // call) - 9.7 KB in all, so that 16 waves fit a CU's 160 KB whatever the allocation granule
    extern __shared__ __attribute__((aligned(16))) float netbuf[];
    float* lds64 = netbuf + 16 * PS;
    SolverLDS* slds_p = SOLVER ? (SolverLDS*)(netbuf + 16 * PS + 64) : nullptr;
    const uint32_t g = g0 + blockIdx.x;
    int lane = threadIdx.x;
    if (g >= E.B) return;
    if (SOLVER && solve_in_flight(E, g)) return;   // the request is still with the solver pool: nothing to do in this launch
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
    R.solve_pending = 0u;
    Slots T;
    T.st = T.sq = T.pk = 0u;
    uint32_t* myblk = E.sim + ((size_t)g * K + (uint32_t)(lane < (int)K ? lane : 0)) * 64;
    if (lane < (int)K) {
        T.st = myblk[GW(sim_state)];
        T.sq = myblk[GW(sim_seq)];
        T.pk = myblk[GW(sim_parked)];
    }
    {
        const uint32_t phase = G32(R, GW(phase));
        if (phase == RAZ_PHASE_DONE || phase == RAZ_PHASE_IDLE || G32(R, GW(error))) {
            if (lane < (int)K) E.nn_active[(size_t)g * K + lane] = 0;
            return;
        }
    }
    raz_net16_zero_planes(netbuf, lane);
    constexpr uint32_t kStageB = 0u, kStageC = 1u, kStageC2 = 2u, kStageD = 4u;  // D outlives an iteration only under a suspended solve
    uint32_t stage = G32(R, GW(par_stage));
    unsigned long long dmask = stage == kStageD ? (unsigned long long)G32(R, GW(par_dmask)) : 0ULL;  // sleepers still to poll in D
    const int lane0 = lane;
    const raz_engine_dev& E0 = E;
    for (uint32_t it = 0; it < iters; ++it) {
        const int lane = fresh_lane<(RAZ_FRESH_K & 1)>(lane0);
#if (RAZ_FRESH_K >> 3) & 1
        const raz_engine_dev& E = fresh_descriptor<1>(E0);
#else
        const raz_engine_dev E = fresh_config<(RAZ_FRESH_K >> 1) & 1>(E0);
#endif
        {
            const uint32_t phase = G32(R, GW(phase));
            if (phase == RAZ_PHASE_DONE || phase == RAZ_PHASE_IDLE || G32(R, GW(error))) break;
        }
        uint32_t nnmask = 0u;
        int budget = (int)K + (((E.cfg.reserved >> 12) & 0xf) ? (int)((E.cfg.reserved >> 12) & 0xf) : kInnerMax);
        for (;;) {
            const int lane = fresh_lane<(RAZ_FRESH_K >> 2) & 1>(lane0);
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
                        stage = kStageB;  // the round is complete: the next iteration starts with B
                        break;
                    }
                    const uint32_t pl = G32(R, GW(player)) - 1;
                    const bool sl = lane < (int)K && T.st == RAZ_SIM_WAIT_EXPAND;
                    uint32_t tg = 0u;
                    if (sl) tg = node_hdr(node_ptr(E, g, T.pk))->tag;
                    dmask = __ballot(sl && !((tg >> (6 + pl)) & 1u)) & kmask;
                    stage = kStageD;
                    continue;
                }
                if (budget <= 0) break;  // the fill goes on at the next iteration, without a B in between
                --budget;
                j = __ffsll((long long)(~busy & kmask)) - 1;
            }
            // ---- at most one slot load, one descent, one return
            bool back = resume;
            if (resume || wake || suspended) slot_load(E, R, g, (uint32_t)j, lane, resume);
            if (!resume) {
                select_leaf<SOLVER, true>(E, R, g, lane, slds_p, g * K + (uint32_t)j,
                                          suspended ? G32(R, GW(leaf_node)) : (wake ? lane_u32(T.pk, j) : G32(R, GW(root_node))),
                                          (wake || suspended) ? (int)G32(R, GW(depth)) : 0, wake && !suspended,
                                          suspended ? (int)G32(R, GW(leaf_action)) - 1 : -1);
                const uint32_t kind = G32(R, GW(leaf_kind));
                if (SOLVER && kind == RAZ_LEAF_SOLVE_PENDING) {  // out of solver budget: the slot keeps the descent, the launch is over for the game
                    T.st = writelane_r(T.st, RAZ_SIM_SOLVING, j, lane);
                    slot_store(E, R, g, (uint32_t)j, lane);
                    S32(R, GW(leaf_kind), RAZ_LEAF_NONE);
                    break;
                }
                if (kind == RAZ_LEAF_TERMINAL || kind == RAZ_LEAF_SOLVED) {
                    back = true;
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
                backup_leaf<true>(E, R, g, G32(R, GW(player)) - 1, lane, lds64);
                T.st = writelane_r(T.st, RAZ_SIM_FREE, j, lane);
            }
        }
        if (stage == kStageD && !(SOLVER && R.solve_pending)) stage = kStageC2;   // (what the classic kernel stores at the end of a launch)
        // ---- the net batch of this iteration: the leaves queued above, evaluated by this wave (their answers go where the net
        // kernel would have put them; the B phase of the next iteration picks them up with slot_load)
        wave_sync();
        for (uint32_t m = nnmask; m; m &= m - 1) {
            const int lane = fresh_lane<(RAZ_FRESH_K >> 2) & 1>(lane0);
            const uint32_t jj = (uint32_t)__ffs((int)m) - 1u;
            const size_t gi = (size_t)g * K + jj;
            const raz_bb own = uni((raz_bb)E.nn_own[gi]), enemy = uni((raz_bb)E.nn_enemy[gi]);
            float pol, val;
            raz_net16_forward_in_wave(net_w, net_R, net_V, own, enemy, netbuf, lane, pol, val);
            E.nn_policy[gi * 64 + lane] = pol;
            if (lane == 0) E.nn_value[gi] = val;
        }
        wave_sync();
        if (SOLVER && R.solve_pending) break;   // a solve (the root's, or one inside a simulation) waits for the next launch's budget
    }
    if (SOLVER && stage == kStageD) S32(R, GW(par_dmask), (uint32_t)dmask);
    S32(R, GW(par_stage), stage);
    lane = fresh_lane<(RAZ_FRESH_K & 1)>(lane);
    gw[lane] = R.cw;
    myblk = E.sim + ((size_t)g * K + (uint32_t)(lane < (int)K ? lane : 0)) * 64;
    if (lane < (int)K) {
        myblk[GW(sim_state)] = T.st;
        myblk[GW(sim_seq)] = T.sq;
        myblk[GW(sim_parked)] = T.pk;
        E.nn_active[(size_t)g * K + lane] = 0;
    }
}

}  // namespace

// `n_steps` simulation steps of the whole batch in ceil(n_steps / 32) launches on stream s (raz_engine_step, reserved bit 4)
int raz_launch_tree_net(const raz_engine_dev& d, bool solver, uint32_t n_steps, const float* W, int R, int V, hipStream_t s) {
    constexpr uint32_t kFusedIters = RAZ_FUSED_ITERS;
    // behind the plane buffer: the heads' scratch of a forward, shared in time with backup_leaf's 64 floats and - solver forms only - the
    // scalar end-game search's frames (704 B); 16 waves per CU x 9.7 KB either way
    const int between = 64 + (solver ? (int)(sizeof(SolverLDS) / sizeof(float)) : 0);
    const int head = 192 + V > between ? 192 + V : between;
    const size_t shm = ((size_t)16 * PS + head) * sizeof(float);
    int rc = RAZ_OK;
    while (n_steps && rc == RAZ_OK) {
        const uint32_t it = n_steps < kFusedIters ? n_steps : kFusedIters;
        if (d.par && solver)
            hipLaunchKernelGGL(k_tree_par_net<true>, dim3(d.B), dim3(64), shm, s, d, 0u, d.B, it, W, R, V);
        else if (d.par)
            hipLaunchKernelGGL(k_tree_par_net<false>, dim3(d.B), dim3(64), shm, s, d, 0u, d.B, it, W, R, V);
        else if (solver)
            hipLaunchKernelGGL(k_tree_net<true>, dim3(d.B), dim3(64), shm, s, d, 0u, d.B, it, W, R, V);
        else
            hipLaunchKernelGGL(k_tree_net<false>, dim3(d.B), dim3(64), shm, s, d, 0u, d.B, it, W, R, V);
        rc = raz_check_launch("raz_engine_step: k_tree_net");
        n_steps -= it;
    }
    return rc;
}
