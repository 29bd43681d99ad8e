"""Host-side driver objects over the C ABI (include/raz.h): DeviceNet (raz_net_*) and
SelfPlayEngine (raz_engine_*).  torch is used only to own HBM buffers and the HIP stream.

SelfPlayEngine replaces, for a batch of concurrent games, the reference's
SelfPlayWorker.start_game loop (worker/self_play.py:139-175) with two ReversiPlayers per game
(agent/player.py); `records()` returns, per game, exactly the per-ply facts the reference worker
keeps (ReversiPlayer.moves rows are derived from them in worker/self_play.py of this package).
"""
import ctypes
import os

import numpy as np

from . import _native as N
from ._native import lib, check

NODE_POOL_BYTES_PER_NODE = 232   # csrc/raz_engine.h RAZ_NODE_DEFAULT_BYTES: a game's default pool budget per node of nodes_per_game
NODE_MAX_BYTES = 704             # RAZ_NODE_MAX_BYTES: 40 B header + 20 B x 33 legal moves
WHOLE_GAME_BYTES_PER_NODE = 320  # byte budget per node of a pool sized for whole games (never pruned)

PLY_HEADER = np.dtype([("own", "<u8"), ("enemy", "<u8"), ("n", "<f8"), ("q", "<f8"), ("action", "i1"),
                       ("player", "u1"), ("turn", "u1"), ("has_row", "u1"), ("sims", "<u4"),
                       ("loops", "<u4"), ("flags", "<u4")])
assert PLY_HEADER.itemsize == 48


def _stream():
    import torch
    return torch.cuda.current_stream().cuda_stream


class DeviceNet:
    """The policy/value net resident in HBM (agent/api.py ReversiModelAPI role, device side)."""

    def __init__(self, blob: bytes, device="cuda:0", force_valu_kernel=False, kernel=None):
        """kernel: None / "f32" = the exact-f32 kernels chosen by shape (raznet-forward-v1, bit-identical to the CPU oracle);
        "f16x3" = raznet-forward-v2 for filters % 128 == 0: the 3x3 trunk on the f16 matrix cores with split operands, within
        1e-5 of the fp32 graph (include/raz.h raz_net_range_check); "auto" = "f16x3" where supported, else "f32".
        Test variants: "valu" = k_net_wave; "mfma_wave" = one single-wave workgroup per position."""
        import torch
        import struct
        magic, ver, F, R, V = struct.unpack_from("<5i", blob, 0)
        nbytes = lib.raz_net_weight_bytes(F, R, V)
        if nbytes == 0:
            raise ValueError(f"unsupported net shape F={F} R={R} V={V}")
        self.device = torch.device(device)
        self.filters, self.res_layers, self.value_fc = F, R, V
        self._weights = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        self.c = N.RazNet()
        kernel = "valu" if force_valu_kernel else kernel
        if kernel == "auto":
            kernel = "f16x3" if (F >= 128 and F % 128 == 0) else "f32"
        self.kernel_name = {None: "f32", "f32": "f32", "f16x3": "f16x3 split-operand MFMA trunk"}.get(kernel, kernel)
        self.c.reserved = {None: 0, "f32": 0, "valu": 1, "mfma_wave": 2, "f16x3": 4}[kernel]
        with torch.cuda.device(self.device):
            check(lib.raz_net_load(ctypes.byref(self.c), blob, len(blob), self._weights.data_ptr(), nbytes,
                                   _stream()), "raz_net_load")
        self._scratch = None

    def scratch(self, n):
        import torch
        need = lib.raz_net_scratch_bytes(self.filters, self.value_fc, n)
        if need and (self._scratch is None or self._scratch.numel() < need):
            self._scratch = torch.empty(need, dtype=torch.uint8, device=self.device)
        return (self._scratch.data_ptr(), self._scratch.numel()) if need else (None, 0)

    def range_ok(self):
        """False when a split-f16 activation left the f16 range since the net was loaded (raz_net_range_check)."""
        import torch
        v = ctypes.c_int(0)
        with torch.cuda.device(self.device):
            check(lib.raz_net_range_check(ctypes.byref(self.c), ctypes.byref(v), _stream()), "raz_net_range_check")
        return v.value == 0

    def range_stats(self):
        """(in range, rows repaired): rows of split-f16 forwards since the net was loaded whose activations left the f16 range and
        were evaluated on the exact-f32 chains instead (include/raz.h raz_net_range_stats)."""
        import torch
        v, r = ctypes.c_int(0), ctypes.c_ulonglong(0)
        with torch.cuda.device(self.device):
            check(lib.raz_net_range_stats(ctypes.byref(self.c), ctypes.byref(v), ctypes.byref(r), _stream()), "raz_net_range_stats")
        return v.value == 0, int(r.value)

    def predict_bitboards(self, own, enemy, active=None):
        """own/enemy: int64 device tensors (side to move's view).  -> (policy (n,64), value (n,)).
        active (uint8 device tensor, optional): positions with 0 are skipped, their outputs stay 0."""
        import torch
        n = own.numel()
        alloc = torch.zeros if active is not None else torch.empty
        policy = alloc((n, 64), dtype=torch.float32, device=self.device)
        value = alloc((n,), dtype=torch.float32, device=self.device)
        sp, sb = self.scratch(n)
        with torch.cuda.device(self.device):
            check(lib.raz_net_forward(ctypes.byref(self.c), own.data_ptr(), enemy.data_ptr(),
                                      active.data_ptr() if active is not None else None,
                                      policy.data_ptr(), value.data_ptr(), n, sp, sb, _stream()), "raz_net_forward")
        return policy, value


def _tuning_env(name):
    """A tuning knob from the environment - honoured ONLY under RAZ_TUNING=1 (GPU tuning sessions, tools/sessions/): a stray variable in a
    production environment must not reach an engine's configuration."""
    return int(os.environ.get(name, "0") or 0) if os.environ.get("RAZ_TUNING") == "1" else 0


CONTINUOUS_SOLVER_POOL_EVERY = 3   # play_continuous: tree launches per round of the solver pool (tools/sessions/r5_s19.sh: 2: 27.2, 3: 30.3, 4: 30.0, 6: 28.7 M)


def engine_config_from(config, n_games, seed, nodes_per_game, max_plies=72, mirror_updates=None,
                       record_root_w=False, phase_profile=False, single_stream=False, parts=0, inner_max=0, solver_memo_slots=1 << 16,
                       force_slot_kernel=False, pool_bytes_per_game=0, fused=False, solver_budget=0, solver_pool_waves=0, solver_pool_every=0):
    """raz_engine_config from a Config-like object with `.play` / `.play_data` (reference names).
    play.parallel_search_num (config.py:142): 1 = the reference's reproducible mode (k_tree); 2..16 =
    that many simulations in flight per game on the raz-sched-v1 schedule (k_tree_par).
    solver_budget: iterations a worker lane of the end-game solver's pool runs between two tree launches (rounded up to 64; 0 = the library's default, 96);
    solver_pool_waves: worker wavefronts of that pool (0 = one per four games, at most 1280); solver_pool_every: tree launches per round
    of the pool (0 = the library's default, 1 = every step waits for the round, n > 1 = the round runs beside the next n - 1 steps on its
    own stream).  None of them changes a result."""
    p = config.play
    par = int(getattr(p, "parallel_search_num", 1) or 1)
    if not 1 <= par <= 16:
        raise ValueError("parallel_search_num must be 1..16 (prediction_queue_size, config.py:141)")
    # agent/player.py:411: temperature = min(exp(1 - (turn / policy_decay_turn) ** policy_decay_power), 1) is 1 - i.e. inert - for every
    # turn <= policy_decay_turn when the power is >= 0 (config.py:150 "not used": 60, 3).  Anything else would change the search's
    # priors; the engine does not implement it and says so instead of ignoring the value.
    pdt, pdp = getattr(p, "policy_decay_turn", 60), getattr(p, "policy_decay_power", 3)
    if pdt is not None and pdp is not None and (float(pdt) < 60 or float(pdp) < 0):
        raise ValueError(f"policy_decay_turn={pdt} / policy_decay_power={pdp} would decay the policy inside a game (agent/player.py:411); "
                         "only the inert setting (turn >= 60, power >= 0) is supported")
    ust = int(getattr(p, "use_solver_turn", 0) or 0)
    usts = int(getattr(p, "use_solver_turn_in_simulation", 0) or 0)
    share = bool(p.share_mtcs_info_in_self_play)
    mirror = share if mirror_updates is None else bool(mirror_updates)
    slots = 16
    while slots < 2 * nodes_per_game:
        slots *= 2
    c = N.RazEngineConfig(
        thinking_loop=p.thinking_loop, required_visit_to_decide_action=p.required_visit_to_decide_action,
        start_rethinking_turn=p.start_rethinking_turn, change_tau_turn=p.change_tau_turn,
        virtual_loss=p.virtual_loss, allowed_resign_turn=p.allowed_resign_turn,
        has_resign_threshold=int(p.resign_threshold is not None), share_mtcs_info=int(share),
        mirror_updates=int(mirror), record_root_w=int(record_root_w), c_puct=float(p.c_puct),
        noise_eps=float(p.noise_eps), dirichlet_alpha=float(p.dirichlet_alpha),
        resign_threshold=float(p.resign_threshold if p.resign_threshold is not None else 0.0),
        disable_resignation_rate=float(p.disable_resignation_rate), n_games=n_games,
        nodes_per_game=nodes_per_game, table_slots=slots, max_plies=max_plies, seed=seed,
        reserved=(1 if phase_profile else 0) | (2 if single_stream else 0) | (8 if force_slot_kernel else 0)
        | (16 if fused else 0) | ((parts & 0xf) << 8) | ((inner_max & 0xf) << 12) | ((min(255, (int(solver_budget) + 63) // 64) & 0xff) << 16) | ((int(solver_pool_every or _tuning_env("RAZ_SOLVER_POOL_EVERY")) & 0xf) << 24),
        use_solver_turn=ust, use_solver_turn_in_simulation=usts,
        solver_memo_slots=(solver_memo_slots if (ust or usts) else 0), parallel_search_num=par,
        pool_bytes_per_game=int(pool_bytes_per_game or 0), solver_pool_waves=int(solver_pool_waves or 0))
    return c


class SelfPlayEngine:
    def __init__(self, config, net: DeviceNet, n_games, seed=0, nodes_per_game=None, sims_hint=None,
                 max_plies=72, mirror_updates=None, record_root_w=False, phase_profile=False, single_stream=False, parts=0, inner_max=0,
                 force_slot_kernel=False, leaf_cache_log2=None, leaf_cache_max_discs=0, pool_bytes_per_game=0, fused=False, solver_budget=0,
                 solver_pool_waves=0, solver_pool_every=0):
        """fused: 16-filter nets - tree and net in ONE kernel, every game's wave evaluating its own leaves (csrc/raz_engine_fused.hip:
        k_tree_net, k_tree_par_net for parallel_search_num > 1; the same results bit for bit, +40 % on BASELINE configs[1]; what
        BatchedSelfPlayWorker selects for 16-filter nets).  No evaluation cache in that form.
        nodes_per_game: most tree nodes a game's pool may hold; pool_bytes_per_game: its bytes (0 = nodes_per_game x 232 + 64 x 704:
        nodes are compact - 40 B + 20 B per legal move, ~212 B on average - include/raz.h).
        leaf_cache_log2: attach a cross-game evaluation cache of 2**leaf_cache_log2 entries (320 B each; include/raz.h
        raz_engine_set_leaf_cache): repeated positions are served from it, bit-identically.  None: no cache.
        leaf_cache_max_discs: cache only positions with at most that many discs (0 = all)."""
        import torch
        self.net = net
        self.device = net.device
        self.n_games = n_games
        if inner_max == 0 and net.filters >= 128 and int(getattr(config.play, "parallel_search_num", 1) or 1) <= 1:
            # a wide net's forward dwarfs the tree kernel: let a game whose simulations end on finished positions (no net
            # evaluation needed) keep simulating inside the launch until it has a leaf for the batch, instead of idling a
            # batch row (results do not depend on this budget)
            inner_max = 8
        if nodes_per_game is None:
            s = sims_hint or config.play.simulation_num_per_move
            loops = max(1, config.play.thinking_loop)
            share = bool(config.play.share_mtcs_info_in_self_play)
            mirror = share if mirror_updates is None else bool(mirror_updates)
            # every simulation adds at most one node (two with mirror keys); ~62 searched plies
            nodes_per_game = (s * loops * 62 + 128) * (2 if mirror else 1)
            if not pool_bytes_per_game:
                # a pool that is never pruned must hold the whole game whatever its mobility: 320 B per node (14 legal moves
                # on average; a game's average is ~8.5) instead of the 232 B default of pruned pools, capped at the link range
                want = nodes_per_game * WHOLE_GAME_BYTES_PER_NODE + 64 * NODE_MAX_BYTES
                pool_bytes_per_game = min(want, 255 << 20)
                if pool_bytes_per_game < want:
                    # the link range (25-bit offsets in 8-byte units) caps a pool at 256 MB: keep the node count consistent with the
                    # bytes that are there, so that the directory / table are sized for what can exist and an overflow shows up as
                    # "pool full" at the count the constructor states, not deep into a run
                    nodes_per_game = (pool_bytes_per_game - 64 * NODE_MAX_BYTES) // WHOLE_GAME_BYTES_PER_NODE
                    import warnings
                    warnings.warn(f"never-pruned node pool capped at 255 MB per game: {nodes_per_game} nodes instead of the {want // WHOLE_GAME_BYTES_PER_NODE} "
                                  "a whole game at this simulations x thinking_loop could create", RuntimeWarning)
        self.cfg = engine_config_from(config, n_games, seed, nodes_per_game, max_plies, mirror_updates,
                                      record_root_w, phase_profile, single_stream, parts, inner_max,
                                      force_slot_kernel=force_slot_kernel, pool_bytes_per_game=pool_bytes_per_game, fused=fused,
                                      solver_budget=solver_budget, solver_pool_waves=solver_pool_waves, solver_pool_every=solver_pool_every)
        self.fused = bool(fused)
        self.pool_bytes = int(pool_bytes_per_game) or (int(nodes_per_game) * NODE_POOL_BYTES_PER_NODE + 64 * NODE_MAX_BYTES)
        self.slots = int(self.cfg.parallel_search_num) or 1   # simulation slots (leaf-exchange rows) per game
        # most nodes one step can add to a game's pool: k_tree completes <= inner_max (default 2) simulations,
        # each adding a leaf and its mirror; k_tree_par starts <= slots + inner_max simulations, wakes <= slots
        # sleepers (a leaf node each) and finishes <= slots expansions (a mirror node each)
        self.nodes_per_step = 2 * (inner_max or 2) if (self.slots == 1 and not force_slot_kernel) \
            else 3 * self.slots + (inner_max or 2)
        nbytes = lib.raz_engine_workspace_bytes(ctypes.byref(self.cfg))
        if nbytes == 0:
            raise ValueError("invalid engine config: " + N.last_error())
        self.workspace_bytes = nbytes
        self._ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.device)
        base = (self._ws.data_ptr() + 255) // 256 * 256
        # the engine's own net scratch: DeviceNet.scratch() may re-allocate its buffer for a later, larger predict call
        need = lib.raz_net_scratch_bytes(net.filters, net.value_fc, n_games * self.slots)
        self._scratch = torch.empty(need, dtype=torch.uint8, device=self.device) if need else None
        sp, sb = (self._scratch.data_ptr(), need) if need else (None, 0)
        self._h = ctypes.c_void_p()
        check(lib.raz_engine_create(ctypes.byref(self.cfg), ctypes.byref(net.c), base, nbytes, sp, sb,
                                    ctypes.byref(self._h)), "raz_engine_create")
        self.record_root_w = record_root_w
        self.max_plies = max_plies
        self._cache = None
        if leaf_cache_log2:
            self.attach_leaf_cache(leaf_cache_log2, leaf_cache_max_discs)

    def attach_leaf_cache(self, log2_entries, max_discs=0):
        """(Re-)attach a cleared evaluation cache; call again after the net's weights changed."""
        import torch
        need = lib.raz_leaf_cache_bytes(log2_entries, self.n_games * self.slots)
        if need == 0:
            raise ValueError("leaf_cache_log2 must be 10..28")
        if self._cache is None or self._cache.numel() < need + 256:
            self._cache = torch.empty(need + 256, dtype=torch.uint8, device=self.device)
        base = (self._cache.data_ptr() + 255) // 256 * 256
        with torch.cuda.device(self.device):
            check(lib.raz_engine_set_leaf_cache(self._h, base, need, log2_entries, max_discs, _stream()), "raz_engine_set_leaf_cache")

    def leaf_cache_stats(self):
        """{hits, in_batch_duplicates, evaluated, no_room} since the cache was attached (zeros without a cache)."""
        import torch
        out = (ctypes.c_uint64 * 4)()
        with torch.cuda.device(self.device):
            check(lib.raz_engine_leaf_cache_stats(self._h, out, _stream()), "raz_engine_leaf_cache_stats")
        return {"hits": out[0], "in_batch_duplicates": out[1], "evaluated": out[2], "no_room": out[3]}

    def solver_stats(self):
        """The end-game solver pool's counters since start() (include/raz.h raz_engine_solver_stats)."""
        out = (ctypes.c_uint64 * 15)()
        check(lib.raz_engine_solver_stats(self._h, out, _stream()), "raz_engine_solver_stats")
        tk_slow, tk_pop, tk_all = (int(x) for x in out[:3])
        out = out[3:]
        self._solver_ticks = {"slow_phase_share_of_worker_time": tk_slow / tk_all if tk_all else None, "pops_share": tk_pop / tk_all if tk_all else None,
                              "ticks_per_wave_iteration": None}
        req, ans, rounds, busy, wave_iters, done, skipped, wave_launches, posted, posted_max, listed, listed_max = (int(x) for x in out)
        self._solver_ticks["ticks_per_wave_iteration"] = tk_all / wave_iters if wave_iters else None
        return {"ticks": self._solver_ticks, "requests_posted": posted, "most_requests_of_one_game": posted_max, "rounds_listed_all_games": listed, "most_rounds_listed_of_one_game": listed_max,
                "solves": req, "answers": ans, "pool_rounds_per_answer": rounds / ans if ans else None,
                "lane_utilisation": busy / (64.0 * wave_iters) if wave_iters else None, "busy_lane_iterations": busy,
                "wave_iterations": wave_iters, "subtrees_finished": done, "subtrees_skipped": skipped, "worker_wave_launches": wave_launches}

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            lib.raz_engine_destroy(h)
            self._h = None

    def start(self, first_game_id, sims_per_move, n_active=None):
        import torch
        sims = np.full(self.n_games, sims_per_move, dtype=np.uint32) if np.isscalar(sims_per_move) \
            else np.ascontiguousarray(sims_per_move, dtype=np.uint32)
        assert sims.size == self.n_games
        with torch.cuda.device(self.device):
            check(lib.raz_engine_start(self._h, first_game_id, sims.ctypes.data,
                                       self.n_games if n_active is None else n_active, _stream()), "raz_engine_start")
        self.n_active = self.n_games if n_active is None else n_active

    def next_game(self, first_game_id, sims_per_move, n_active=None):
        """The next game of every slot on the slot's tree (reset_mtcs_info_per_game > 1, include/raz.h)."""
        import torch
        sims = np.full(self.n_games, sims_per_move, dtype=np.uint32) if np.isscalar(sims_per_move) \
            else np.ascontiguousarray(sims_per_move, dtype=np.uint32)
        assert sims.size == self.n_games
        with torch.cuda.device(self.device):
            check(lib.raz_engine_next_game(self._h, first_game_id, sims.ctypes.data,
                                           self.n_games if n_active is None else n_active, _stream()), "raz_engine_next_game")
        self.n_active = self.n_games if n_active is None else n_active

    def step(self, n=1):
        import torch
        with torch.cuda.device(self.device):
            check(lib.raz_engine_step(self._h, n, _stream()), "raz_engine_step")

    def set_parts(self, parts):
        import torch
        torch.cuda.synchronize(self.device)
        check(lib.raz_engine_set_parts(self._h, parts), "raz_engine_set_parts")

    def set_solver_pool_every(self, n):
        """Tree launches per round of the end-game solver's pool (include/raz.h raz_engine_set_solver_pool_every): 1 = every step
        waits for the round, n > 1 = the round runs beside the next n - 1 steps, 0 = the library's default.  Between steps only."""
        import torch
        torch.cuda.synchronize(self.device)
        check(lib.raz_engine_set_solver_pool_every(self._h, int(n)), "raz_engine_set_solver_pool_every")

    def step_timed(self, n=1):
        """step(n) with HIP events around each kernel; returns (tree_ms, net_ms) summed over n."""
        import torch
        a, b = ctypes.c_double(0.0), ctypes.c_double(0.0)
        with torch.cuda.device(self.device):
            check(lib.raz_engine_step_timed(self._h, n, ctypes.byref(a), ctypes.byref(b), _stream()),
                  "raz_engine_step_timed")
        return a.value, b.value

    def phase_profile(self):
        """Per-phase shader-clock ticks summed over games (engine created with phase_profile=True)."""
        import torch
        ptr = lib.raz_engine_device_ptr(self._h, 5)
        off = ptr - self._ws.data_ptr()
        torch.cuda.synchronize(self.device)
        a = self._ws[off:off + self.n_games * 64].cpu().numpy().view(np.uint64).reshape(self.n_games, 8)
        names = ["backup", "controller", "select", "root_noise", "first_arrival_probe", "active_launches",
                 "node_load_wait", "expand_part_of_backup"]
        return {n: int(a[:, i].sum()) for i, n in enumerate(names)}

    def leaf_exchange(self):
        """Views (no copy) of the leaf exchange between the tree kernels and the net as the LAST step left it (include/raz.h
        raz_engine_device_ptr 10-14; the reference's prediction queue, agent/player.py:329-346): {"active" u8 [rows], "own" / "enemy"
        i64 [rows] (as shown to the net), "policy" f32 [rows, 64], "value" f32 [rows]}, rows = n_games x parallel_search_num."""
        import torch
        rows = self.n_games * self.slots
        torch.cuda.synchronize(self.device)

        def view(which, nbytes, dtype):
            off = lib.raz_engine_device_ptr(self._h, which) - self._ws.data_ptr()
            return self._ws[off:off + nbytes].view(dtype)
        return {"active": view(10, rows, torch.uint8), "own": view(11, rows * 8, torch.int64), "enemy": view(12, rows * 8, torch.int64),
                "policy": view(13, rows * 256, torch.float32).view(rows, 64), "value": view(14, rows * 4, torch.float32)}

    def stats(self):
        import torch
        st = N.RazEngineStats()
        with torch.cuda.device(self.device):
            check(lib.raz_engine_stats_sync(self._h, ctypes.byref(st), _stream()), "raz_engine_stats_sync")
        if st.error_flags:
            raise RuntimeError(f"engine error flags {st.error_flags:#x} (1 node pool full, 2 table full, "
                               f"4 records full, 8 path overflow): enlarge nodes_per_game/max_plies")
        return {"finished_games": st.finished_games, "total_sims": st.total_sims, "nn_leaves": st.nn_leaves,
                "selections": st.selections, "max_pool_used": st.max_pool_used, "idle_or_done": st.idle_or_done,
                "max_pool_bytes": st.max_pool_bytes}

    def pool_nearly_full(self, st, steps):
        """True when the fullest running game's pool could overflow within `steps` more steps: a step adds at most
        nodes_per_step nodes of at most 704 B to a game (a pool is full when EITHER its node count reaches nodes_per_game
        or its bytes reach pool_bytes_per_game)."""
        n = self.nodes_per_step * steps + 64
        return st["max_pool_used"] + n > int(self.cfg.nodes_per_game) or st["max_pool_bytes"] + n * NODE_MAX_BYTES > self.pool_bytes

    def set_position(self, slot, black, white, player, sims, enable_resign=True, one_move=True):
        """Arm one move for `slot` on an arbitrary position, keeping the slot's tree (include/raz.h)."""
        import torch
        with torch.cuda.device(self.device):
            check(lib.raz_engine_set_position(self._h, slot, black, white, player, sims, int(enable_resign),
                                              int(one_move), _stream()), "raz_engine_set_position")

    def set_positions(self, first_slot, black, white, player, sims, enable_resign=True, one_move=True):
        """set_position for len(black) consecutive slots in one launch; black / white: int64 device tensors, player: uint8."""
        import torch
        assert black.is_contiguous() and white.is_contiguous() and player.is_contiguous() and player.dtype == torch.uint8
        with torch.cuda.device(self.device):
            check(lib.raz_engine_set_positions(self._h, first_slot, black.numel(), black.data_ptr(), white.data_ptr(), player.data_ptr(),
                                               sims, int(enable_resign), int(one_move), _stream()), "raz_engine_set_positions")

    def stop_thinking(self, slot):
        import torch
        with torch.cuda.device(self.device):
            check(lib.raz_engine_stop_thinking(self._h, slot, _stream()), "raz_engine_stop_thinking")

    def adopt_tree(self, slot, player_index):
        import torch
        with torch.cuda.device(self.device):
            check(lib.raz_engine_adopt_tree(self._h, slot, player_index, _stream()), "raz_engine_adopt_tree")

    def read_node(self, slot, black, white, next_player=1, owner=0):
        """(found, W f64[64], N u32[64], P f32[64]) of one key of one slot's tree (include/raz.h)."""
        import torch
        w = np.zeros(64, dtype=np.float64)
        n = np.zeros(64, dtype=np.uint32)
        p = np.zeros(64, dtype=np.float32)
        found = ctypes.c_int(0)
        with torch.cuda.device(self.device):
            check(lib.raz_engine_read_node(self._h, slot, black, white, next_player, owner, w.ctypes.data,
                                           n.ctypes.data, p.ctypes.data, ctypes.byref(found), _stream()),
                  "raz_engine_read_node")
        return bool(found.value), w, n, p

    def set_resign_threshold(self, threshold):
        """config.play.resign_threshold changed (worker/self_play.py:250-260); None = no resignation rule."""
        check(lib.raz_engine_set_resign_threshold(self._h, int(threshold is not None),
                                                  float(threshold if threshold is not None else 0.0)),
              "raz_engine_set_resign_threshold")

    def pack_records(self, first_slot=0, n_slots=None, plies=None):
        """The records of slots [first_slot, first_slot + n_slots) cut to `plies` plies (None: the largest n_plies
        among them), as dense DEVICE tensors (uint8 views) a collective can move as they are (include/raz.h
        raz_engine_pack_records): {"headers": [n, plies, 48] u8, "root_n": [n, plies, 64] i32, "summary": [n, 32] u8}."""
        import torch
        n = self.n_games - first_slot if n_slots is None else n_slots
        with torch.cuda.device(self.device):
            if plies is None:
                mp = ctypes.c_uint32(0)
                check(lib.raz_engine_records_extent(self._h, first_slot, n, ctypes.byref(mp), _stream()),
                      "raz_engine_records_extent")
                plies = max(1, mp.value)
            hdr = torch.empty((n, plies, 48), dtype=torch.uint8, device=self.device)
            rn = torch.empty((n, plies, 64), dtype=torch.int32, device=self.device)
            sm = torch.empty((n, 32), dtype=torch.uint8, device=self.device)
            check(lib.raz_engine_pack_records(self._h, first_slot, n, plies, hdr.data_ptr(), rn.data_ptr(),
                                              sm.data_ptr(), _stream()), "raz_engine_pack_records")
        return {"headers": hdr, "root_n": rn, "summary": sm}

    # ---- continuous batching (include/raz.h raz_engine_harvest) -------------------------------------------------
    def new_outbox(self, first_game_id, n_games):
        """Device arrays that receive finished games in id order (row = game id - first_game_id)."""
        import torch
        mp = self.max_plies
        return {"first": first_game_id, "n": n_games,
                "headers": torch.zeros((n_games, mp, 48), dtype=torch.uint8, device=self.device),
                "root_n": torch.zeros((n_games, mp, 64), dtype=torch.int32, device=self.device),
                "summary": torch.zeros((n_games, 32), dtype=torch.uint8, device=self.device),
                "done": torch.zeros(n_games, dtype=torch.uint8, device=self.device)}

    def harvest(self, outbox, next_game_id, sims_per_move, resign_threshold=None):
        """Move every finished game into `outbox` and restart the freed slots on ids next_game_id, next_game_id + 1, ...
        (len(sims_per_move) of them at most).  resign_threshold: None = the engine's run-time value; else one value per
        new id (None entries / NaN = no resignation rule).  Returns (harvested, restarted, skipped, playing)."""
        import torch
        sims = np.ascontiguousarray(sims_per_move, dtype=np.uint32)
        thr = None
        if resign_threshold is not None:
            thr = np.array([np.nan if t is None else float(t) for t in resign_threshold], dtype=np.float64)
            assert thr.size == sims.size
        res = N.RazHarvestResult()
        with torch.cuda.device(self.device):
            check(lib.raz_engine_harvest(self._h, next_game_id, sims.size, sims.ctypes.data if sims.size else None,
                                         thr.ctypes.data if thr is not None and thr.size else None, outbox["first"], outbox["n"],
                                         outbox["headers"].data_ptr(), outbox["root_n"].data_ptr(), outbox["summary"].data_ptr(),
                                         outbox["done"].data_ptr(), ctypes.byref(res), _stream()), "raz_engine_harvest")
        return res.harvested, res.restarted, res.skipped, res.playing

    def play_continuous(self, first_game_id, total_games, sims_of, chunk=64, resign_threshold_of=None, max_steps=100_000_000,
                        on_chunk=None):
        """Play global game ids first_game_id .. first_game_id + total_games - 1 with continuous batching: the batch's
        slots are refilled with the next unplayed id as games finish (worker/self_play.py:95-137: a reference worker
        starts its next game the moment one ends), finished games land in an id-ordered device outbox.
        sims_of(id) -> simulations per move of that game (the reference's per-game-index schedule, self_play.py:145); or an array
        with one entry per id of the call (entry i = id first_game_id + i): what the worker passes - a Python call per id and poll was
        3 M calls per 65 536-id block.
        resign_threshold_of(id) (optional) -> the threshold that game is played under.
        on_chunk(steps, games_done, stats, outbox) (optional) is called after every chunk's harvest (progress reporting; the worker
        ships the finished prefix of the outbox to its file writer from it).
        Returns (outbox, stats): outbox = device tensors in id order (pack/gather them as they are, or
        raw_from_packed(...) on the host); stats adds steps, leaf_slot_occupancy and gc_runs."""
        B = self.n_games
        n0 = min(B, total_games)
        # with slots at every stage of a game only some of them wait for the end-game solver at any time: its pool's round runs beside
        # three tree launches instead of holding every step up (mini.yml as shipped: 22.7 M -> 30.3 M sims/s; a value given at
        # construction - solver_pool_every, or RAZ_SOLVER_POOL_EVERY under RAZ_TUNING=1 - stays)
        solver_on = bool(self.cfg.use_solver_turn or self.cfg.use_solver_turn_in_simulation)
        tuned = solver_on and not ((int(self.cfg.reserved) >> 24) & 0xf)
        if tuned:
            self.set_solver_pool_every(CONTINUOUS_SOLVER_POOL_EVERY)
        import time
        t_begin = time.perf_counter()
        try:
            if not callable(sims_of):
                table = np.ascontiguousarray(sims_of, dtype=np.uint32)
                assert table.size == total_games
                sims_at = lambda lo, n: table[lo - first_game_id: lo - first_game_id + n]
            else:
                sims_at = lambda lo, n: np.array([sims_of(i) for i in range(lo, lo + n)], dtype=np.uint32)
            sims0 = np.ones(B, dtype=np.uint32)
            sims0[:n0] = sims_at(first_game_id, n0)
            if resign_threshold_of is not None:
                self.set_resign_threshold(resign_threshold_of(first_game_id))
            self.start(first_game_id, sims0, n_active=n0)
            if resign_threshold_of is not None and len({resign_threshold_of(first_game_id + i) for i in range(n0)}) > 1:
                raise ValueError("the first batch of ids must share one resign threshold (raz_engine_start takes the engine's run-time value)")
            outbox = self.new_outbox(first_game_id, total_games)
            nxt, done, steps, cap = first_game_id + n0, 0, 0, int(self.cfg.nodes_per_game)
            end = first_game_id + total_games
            self.gc_runs = 0
            host = {"steps_and_stats": 0.0, "harvest": 0.0, "before_the_first_step": time.perf_counter() - t_begin}
            while done < total_games:
                t0 = time.perf_counter()
                self.step(chunk)
                steps += chunk
                st = self.stats()
                if self.pool_nearly_full(st, chunk):
                    self.gc(threshold=min(cap // 4, st["max_pool_used"] // 2))
                    self.gc_runs += 1
                t1 = time.perf_counter()
                k = min(B, end - nxt)
                ids = range(nxt, nxt + k)
                h, r, skipped, playing = self.harvest(outbox, nxt, sims_at(nxt, k),
                                                      [resign_threshold_of(i) for i in ids] if resign_threshold_of else None)
                assert skipped == 0
                nxt += r
                done += h
                host["steps_and_stats"] += t1 - t0
                host["harvest"] += time.perf_counter() - t1
                if on_chunk is not None:
                    on_chunk(steps, done, st, outbox)
                if steps >= max_steps:
                    raise RuntimeError("engine did not finish within max_steps")
            st = self.stats()
            host["whole_call"] = time.perf_counter() - t_begin
            st.update(steps=steps, leaf_slot_occupancy=st["nn_leaves"] / max(1, steps * B * self.slots), gc_runs=self.gc_runs,
                      seconds=host)
        finally:
            if tuned:   # (also when a step raised: the engine may be used again by the caller)
                self.set_solver_pool_every(0)
        return outbox, st

    def gc(self, threshold=0):
        """Prune unreachable nodes in every game whose pool holds >= threshold nodes."""
        import torch
        with torch.cuda.device(self.device):
            check(lib.raz_engine_gc(self._h, threshold, _stream()), "raz_engine_gc")

    def run(self, chunk=64, max_steps=10_000_000, allow_gc=True):
        """Step until every active game has finished.  Returns the final stats.  Pools are pruned
        whenever the fullest one could overflow before the next poll (<= nodes_per_step new nodes per step).
        allow_gc=False: never prune (a tree that the NEXT game of the slot will search again must keep its
        early positions: reset_mtcs_info_per_game > 1); a pool that fills up then raises from stats()."""
        steps = 0
        cap = int(self.cfg.nodes_per_game)
        self.gc_runs = 0
        while True:
            self.step(chunk)
            steps += chunk
            st = self.stats()
            if allow_gc and self.pool_nearly_full(st, chunk):
                self.gc(threshold=min(cap // 4, st["max_pool_used"] // 2))
                self.gc_runs += 1
            if st["finished_games"] >= self.n_active:
                st["steps"] = steps
                return st
            if steps >= max_steps:
                raise RuntimeError("engine did not finish within max_steps")

    def read_raw(self):
        import torch
        B, MP = self.n_games, self.max_plies
        hdr = np.zeros((B, MP), dtype=PLY_HEADER)
        root_n = np.zeros((B, MP, 64), dtype=np.uint32)
        root_w = np.zeros((B, MP, 64), dtype=np.float64) if self.record_root_w else None
        n_plies = np.zeros(B, dtype=np.uint32)
        status = np.zeros(B, dtype=np.uint8)
        resigned = np.zeros((B, 2), dtype=np.uint8)
        game_id = np.zeros(B, dtype=np.uint32)
        enable_resign = np.zeros(B, dtype=np.uint8)
        fb = np.zeros(B, dtype=np.uint64)
        fw = np.zeros(B, dtype=np.uint64)
        with torch.cuda.device(self.device):
            check(lib.raz_engine_read_records(
                self._h, hdr.ctypes.data, root_n.ctypes.data, root_w.ctypes.data if root_w is not None else None,
                n_plies.ctypes.data, status.ctypes.data, resigned.ctypes.data, game_id.ctypes.data,
                enable_resign.ctypes.data, fb.ctypes.data, fw.ctypes.data, _stream()), "raz_engine_read_records")
        return dict(headers=hdr, root_n=root_n, root_w=root_w, n_plies=n_plies, status=status, resigned=resigned,
                    game_id=game_id, enable_resign=enable_resign, final_black=fb, final_white=fw)

    def records(self, save_policy_of_tau_1=True, change_tau_turn=None):
        """Per game: (plies, summary) in the same shape oracle.selfplay_game returns."""
        raw = self.read_raw()
        ctt = self.cfg.change_tau_turn if change_tau_turn is None else change_tau_turn
        out = []
        for g in range(self.n_active):
            plies = []
            for i in range(int(raw["n_plies"][g])):
                h = raw["headers"][g, i]
                n = raw["root_n"][g, i].astype(np.float64)
                plies.append({"player": int(h["player"]), "turn": int(h["turn"]), "own": int(h["own"]),
                              "enemy": int(h["enemy"]), "action": int(h["action"]), "has_row": bool(h["has_row"]), "solved": bool(int(h["flags"]) & 1),
                              "sims": int(h["sims"]), "loops": int(h["loops"]), "n": float(h["n"]), "q": float(h["q"]),
                              "root_n": [float(v) for v in n],
                              "root_w": [float(v) for v in raw["root_w"][g, i]] if raw["root_w"] is not None else None,
                              "saved_policy": saved_policy(n, int(h["turn"]), ctt, save_policy_of_tau_1)})
            st = int(raw["status"][g])
            out.append((plies, {"winner": st & 0x0f, "status": st, "plies": len(plies),
                                "game_id": int(raw["game_id"][g]), "enable_resign": int(raw["enable_resign"][g]),
                                "resigned_black": int(raw["resigned"][g, 0]), "resigned_white": int(raw["resigned"][g, 1]),
                                "black": int(raw["final_black"][g]), "white": int(raw["final_white"][g])}))
        return out


GAME_SUMMARY = np.dtype([("final_black", "<u8"), ("final_white", "<u8"), ("game_id", "<u4"), ("n_plies", "<u4"),
                         ("status", "u1"), ("resigned_black", "u1"), ("resigned_white", "u1"), ("enable_resign", "u1"),
                         ("sims", "<u4")])
assert GAME_SUMMARY.itemsize == 32


def raw_from_packed(headers_u8, root_n_i32, summary_u8):
    """Host numpy views of packed record arrays (SelfPlayEngine.pack_records, possibly concatenated over ranks)
    in the dict shape SelfPlayEngine.read_raw returns (minus root_w)."""
    hdr = np.ascontiguousarray(headers_u8).view(PLY_HEADER).reshape(headers_u8.shape[0], headers_u8.shape[1])
    sm = np.ascontiguousarray(summary_u8).view(GAME_SUMMARY).reshape(-1)
    return dict(headers=hdr, root_n=np.ascontiguousarray(root_n_i32).view(np.uint32), root_w=None,
                n_plies=sm["n_plies"].copy(), status=sm["status"].copy(),
                resigned=np.stack([sm["resigned_black"], sm["resigned_white"]], axis=1),
                game_id=sm["game_id"].copy(), enable_resign=sm["enable_resign"].copy(), sims=sm["sims"].copy(),
                final_black=sm["final_black"].copy(), final_white=sm["final_white"].copy())


def saved_policy(root_n, turn, change_tau_turn, save_policy_of_tau_1):
    """The policy stored with a training row (agent/player.py:132, 366-385), from the root visit
    counts, with the reference's own numpy expressions."""
    n = np.asarray(root_n, dtype=np.float64)
    if save_policy_of_tau_1 or turn < change_tau_turn:
        return list(n / np.sum(n))
    ret = np.zeros(64)
    ret[int(np.argmax(n))] = 1
    return list(ret)
