// raz_engine_fused.hip — tree + net in ONE kernel for narrow nets: k_tree_net.
//
// With the F == 16 net (mini.yml) a step of the classic pipeline is two latency-bound launches per slice - k_tree (25 us) and
// k_net_mfma (one wave per position, 13-25 us) - handing leaves and answers over through HBM, and its rate is set by that chain's
// latency and by how many launches per second one host thread and the hardware queues sustain (more than three
// slices or hipGraph replay made it slower).  Here the game's wave evaluates its own leaf: the descent ends with the position
// in registers, raz_net16_forward_in_wave (raz_net_wave.h: matrix-core trunk in one LDS plane buffer of this wave, heads as
// k_net_mfma) returns the policy row and the value into the registers backup_leaf reads, and the loop goes on with the next
// simulation - `iters` of them per launch, the control block and the path staying in registers throughout.  No leaf exchange, no
// second kernel, no slices; one launch per `iters` simulation steps of the whole batch.  4 waves per SIMD (128 VGPRs, 10 KB of
// LDS per wave): 4096 games are resident at once.  Every game performs exactly the operations it performs under k_tree +
// k_net_mfma, in the same order: results are bit-identical (tests/test_engine_fused_emu.py on the wave emulator,
// tests/test_engine_gpu.py).  Opt-in (raz_engine_config.reserved bit 4) for parallel_search_num == 1 without the evaluation
// cache; at the end of a launch the last leaf's answer is parked in nn_policy / nn_value exactly where k_net_mfma would have
// put it, so the two forms can even alternate.
#include <hip/hip_runtime.h>
#include "raz_engine_core.h"
#include "raz_net_wave.h"   // raz_net16_forward_in_wave

namespace {

template <bool SOLVER>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_tree_net(raz_engine_dev E, uint32_t g0, uint32_t count,
                                                                                          uint32_t iters, const float* __restrict__ net_w,
                                                                                          int net_R, int net_V) {
    if (blockIdx.x >= count) return;
    __shared__ float lds64[64];
    __shared__ SolverLDS slds_store;
    extern __shared__ __attribute__((aligned(16))) float netbuf[];
    SolverLDS* slds_p = SOLVER ? &slds_store : nullptr;
    const uint32_t g = g0 + blockIdx.x;
    const int lane = threadIdx.x;
    if (g >= E.B) return;
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
        if (phase == RAZ_PHASE_DONE || phase == RAZ_PHASE_IDLE || G32(R, GW(error))) return;
    }
    raz_net16_zero_planes(netbuf, lane);
    for (uint32_t it = 0; it < iters; ++it) {
        uint32_t phase = G32(R, GW(phase));
        if (phase == RAZ_PHASE_DONE || phase == RAZ_PHASE_IDLE) break;
        if (G32(R, GW(error))) break;
        if (G32(R, GW(leaf_kind)) != RAZ_LEAF_NONE) backup_leaf<false>(E, R, g, G32(R, GW(player)) - 1, lane, lds64);
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
        phase = G32(R, GW(phase));
        if (phase != RAZ_PHASE_SEARCH || (int32_t)G32(R, GW(sims_left)) <= 0 || G32(R, GW(error))) break;
        select_leaf<SOLVER, false>(E, R, g, lane, slds_p, g, 0u, 0, false);
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
        } else if (lk != RAZ_LEAF_TERMINAL && lk != RAZ_LEAF_SOLVED)
            break;
    }
    gw[lane] = R.cw;
    if (R.path_dirty) path_store(E, R, (size_t)g, lane);
    // the answer for a leaf that is still to be backed up waits where the net kernel would have left it
    E.nn_policy[(size_t)g * 64 + lane] = R.pol_raw;
    if (lane == 0) {
        E.nn_value[g] = R.val;
        E.nn_active[g] = 0;
    }
}

}  // namespace

// `n_steps` simulation steps of the whole batch in ceil(n_steps / 32) launches on stream s (raz_engine_step, reserved bit 4)
int raz_launch_tree_net(const raz_engine_dev& d, bool solver, uint32_t n_steps, const float* W, int R, int V, hipStream_t s) {
    constexpr uint32_t kFusedIters = 32;
    const size_t shm = (size_t)raz_net16_lds_floats(V) * sizeof(float);
    int rc = RAZ_OK;
    while (n_steps && rc == RAZ_OK) {
        const uint32_t it = n_steps < kFusedIters ? n_steps : kFusedIters;
        if (solver)
            hipLaunchKernelGGL(k_tree_net<true>, dim3(d.B), dim3(64), shm, s, d, 0u, d.B, it, W, R, V);
        else
            hipLaunchKernelGGL(k_tree_net<false>, dim3(d.B), dim3(64), shm, s, d, 0u, d.B, it, W, R, V);
        rc = raz_check_launch("raz_engine_step: k_tree_net");
        n_steps -= it;
    }
    return rc;
}
