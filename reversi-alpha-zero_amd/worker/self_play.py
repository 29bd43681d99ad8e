"""The `self` worker on the batched engine — drop-in for reversi_zero/worker/self_play.py.

`start(config)` is the reference's entry point (worker/self_play.py:28).  Instead of
multi_process_num Python processes each playing one game at a time, ONE process per GPU keeps
`games_in_flight` games resident on the device (SelfPlayEngine) and emits exactly the files the
reference worker emits:
    data/play_data/play_<timestamp>.json   rows [[own, enemy], policy64, z]   (self_play.py:180-194)
    data/self_play-ggf/self_play-<ts>.ggf  (self_play.py:196-207, MoveHistory :275-299)
    data/.self-play-game-idx               (self_play.py:136-137);  data/.force-sim honoured (:262-267)
so the reference's `opt` / `eval` workers consume them unchanged.

Multi-GPU: launched with torch.distributed (one rank per GPU); rank r plays global game ids
[first + r*B, first + (r+1)*B); results do not depend on the sharding because every random draw is
keyed by the global game id.  The only collective with a payload is a gather to rank 0 (RCCL over xGMI
on GPUs, gloo in the CPU tests): of the finished games' records when rank 0 writes every file
(emission "rank0"), of their 32-byte summaries when a rank's id range is a whole number of files and
every rank writes its own (emission "per_rank"; "auto" picks it where it applies).  Same files either way.
"""
import os
from datetime import datetime, timedelta
from logging import getLogger

import numpy as np

from ..lib import bitboard as bb
from ..lib.data_helper import get_game_data_filenames, write_game_data_to_file
from ..lib.ggf import convert_action_to_move, make_ggf_string

logger = getLogger(__name__)


def read_as_int(filename):
    """lib/file_util.py:4-12."""
    if os.path.exists(filename):
        try:
            with open(filename, "rt") as f:
                return int(str(f.read()).strip())
        except ValueError:
            pass
    return None


def decide_simulation_num_per_move(config, idx):
    """worker/self_play.py:262-272: data/.force-sim overrides the per-game-index schedule."""
    ret = read_as_int(config.resource.force_simulation_num_file)
    if ret:
        return ret
    for min_idx, num in config.play.schedule_of_simulation_num_per_move:
        if idx >= min_idx:
            ret = num
    return ret


def simulation_nums_of_ids(config, first_idx, n):
    """decide_simulation_num_per_move for the ids first_idx .. first_idx + n - 1 as one array.  data/.force-sim is read ONCE for the
    block (the reference looks at it at every game's start: a game here starts when a slot frees up inside a block that is played in
    seconds to minutes, so a changed file takes effect with the next block - and 65 536 stat() calls per block were a quarter of the
    worker's wall time on a 16-filter net)."""
    forced = read_as_int(config.resource.force_simulation_num_file)
    if forced:
        return np.full(n, forced, dtype=np.uint32)
    ids = np.arange(first_idx, first_idx + n, dtype=np.int64)
    out = np.zeros(n, dtype=np.int64)
    seen = np.zeros(n, dtype=bool)
    for min_idx, num in config.play.schedule_of_simulation_num_per_move:   # (the last matching entry wins, as in the reference's loop)
        hit = ids >= min_idx
        out[hit] = num
        seen |= hit
    if not seen.all():
        raise ValueError("schedule_of_simulation_num_per_move has no entry for game index %d" % int(ids[~seen][0]))
    return out.astype(np.uint32)


def symmetric_rows(own, enemy, policy):
    """add_data_to_move_buffer_with_8_symmetries (agent/player.py:166-179): the 8 rows
    [(own, enemy), policy64] of one searched ply, flip in {F,T} x rot_right in 0..3."""
    policy = np.asarray(policy, dtype=np.float64)
    rows = []
    for flip in (False, True):
        for rot in range(4):
            o, e, pol = own, enemy, policy.reshape(8, 8)
            if flip:
                o, e, pol = bb.flip_vertical(o), bb.flip_vertical(e), np.flipud(pol)
            for _ in range(rot):
                o, e = bb.rotate90(o), bb.rotate90(e)
            if rot:
                pol = np.rot90(pol, k=-rot)
            rows.append([(o, e), list(pol.reshape(64))])
    return rows


def rows_of_game(plies, winner):
    """The training rows of one finished game in file order: black.moves + white.moves
    (worker/self_play.py:183), each searched ply contributing its 8 symmetric rows
    (agent/player.py:166-179: flip in {F,T} x rot_right in 0..3; boards by flip_vertical/rotate90,
    policy by np.flipud / np.rot90(k=-rot)), z appended by finish_game (player.py:357-364,
    self_play.py:219-231: black rows get black_win, white rows -black_win)."""
    black_win = 1 if winner == 1 else (-1 if winner == 2 else 0)
    per_player = {1: [], 2: []}
    for p in plies:
        if not p["has_row"]:
            continue
        z = black_win if p["player"] == 1 else -black_win
        per_player[p["player"]] += [row + [z] for row in symmetric_rows(p["own"], p["enemy"], p["saved_policy"])]
    return per_player[1] + per_player[2]


def ggf_moves_of_game(plies):
    """MoveHistory.move (worker/self_play.py:279-296): a pass is recorded as "PA" for the side that
    could not move; resignation adds nothing."""
    moves = []
    for p in plies:
        if p["action"] < 0:
            continue
        if (len(moves) % 2 == 0) != (p["player"] == 1):
            moves.append(convert_action_to_move(None))
        if p.get("solved"):  # action_by_searching returns n=999 (int) and q=np.sign(score) (integer): "10/999"
            moves.append(f"{convert_action_to_move(p['action'])}/{int(p['q']) * 10}/{int(p['n'])}")
        else:
            moves.append(f"{convert_action_to_move(p['action'])}/{p['q'] * 10}/{p['n']}")
    return moves


def drop_draw_uniform(seed, game_id):
    """GAME-purpose second uniform of raz-rng-v1 (the np.random.random() of self_play.py:182)."""
    from .._rng import rng_pair
    return rng_pair(seed, game_id, 3, 0)[1]


def game_rows_json(headers, root_n, n_plies, winner, change_tau_turn, save_policy_of_tau_1, as_bytes=False):
    """The JSON text of one finished game's rows (rows joined by ", ", no enclosing brackets) and their number,
    produced by libraz (csrc/raz_emit.hip) from the engine's raw records: byte for byte what
    json.dumps(rows_of_game(plies, winner)) yields minus the outer "[" "]"."""
    import ctypes
    from .._native import lib, last_error
    n_plies = int(n_plies)
    hdr = np.ascontiguousarray(headers[:n_plies])
    rn = np.ascontiguousarray(root_n[:n_plies], dtype=np.uint32)
    cap = max(1, n_plies) * 8 * 2048
    buf = ctypes.create_string_buffer(cap)
    nrows = ctypes.c_int(0)
    n = lib.raz_emit_game_rows_json(hdr.ctypes.data, rn.ctypes.data, n_plies, int(winner), int(change_tau_turn),
                                    int(bool(save_policy_of_tau_1)), buf, cap, ctypes.byref(nrows))
    if n < 0:
        raise RuntimeError("raz_emit_game_rows_json: " + last_error())
    return (buf.raw[:n] if as_bytes else buf.raw[:n].decode("ascii")), nrows.value


def plies_for_ggf(headers, n_plies):
    """The per-ply facts MoveHistory needs (worker/self_play.py:279-296) from raw ply headers."""
    h = headers[:int(n_plies)]
    return [{"action": a, "player": pl, "solved": bool(f & 1), "q": q, "n": n} for a, pl, f, q, n in
            zip(h["action"].tolist(), h["player"].tolist(), h["flags"].tolist(), h["q"].tolist(), h["n"].tolist())]


RAW_KEYS = ("headers", "root_n", "n_plies", "status", "resigned", "game_id", "enable_resign", "final_black", "final_white")


def gather_raw(raw, rank, world, device=None):
    """The single collective of the path on the engine's own record arrays: every rank's arrays (same shapes
    on all ranks) gathered on rank 0 and concatenated in rank order = global game id order.  Returns None
    on the other ranks."""
    import torch
    import torch.distributed as dist
    backend = dist.get_backend()
    dev = torch.device(device) if device is not None else (
        torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu"))
    out = {}
    for k in RAW_KEYS:
        a = np.ascontiguousarray(raw[k])
        t = torch.from_numpy(a.view(np.uint8).reshape(-1)).to(dev)
        lst = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
        dist.gather(t, lst, dst=0)
        if rank == 0:
            parts = [x.cpu().numpy().view(a.dtype).reshape(a.shape) for x in lst]
            out[k] = np.concatenate(parts, axis=0)
    return out if rank == 0 else None


def gather_packed(packed, rank, world):
    """The single collective of the path, device side: every rank's packed record tensors
    (SelfPlayEngine.pack_records: dense, cut to the games' plies) gathered on rank 0 straight from HBM - RCCL over
    xGMI under the nccl backend; under gloo (CPU test rig) the tensors are staged through host memory - and
    concatenated in rank order = global game id order.  One all_reduce(MAX) of the ply extent first, so that every
    rank packs to the same shape.  Returns (raw dict in read_raw's shape, bytes moved) on rank 0, (None, bytes) elsewhere.
    `packed` is a callable plies -> dict so the pack happens once, at the common extent."""
    import torch
    import torch.distributed as dist
    from ..engine import raw_from_packed
    backend = dist.get_backend()
    first = packed(None)
    dev = first["headers"].device if backend == "nccl" else torch.device("cpu")
    ext = torch.tensor([first["headers"].shape[1]], dtype=torch.int64, device=dev)
    dist.all_reduce(ext, op=dist.ReduceOp.MAX)
    plies = int(ext.item())
    pk = first if plies == first["headers"].shape[1] else packed(plies)
    out, moved = {}, 0
    for k in ("headers", "root_n", "summary"):
        t = pk[k].contiguous().to(dev)
        lst = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
        dist.gather(t, lst, dst=0)
        moved += t.numel() * t.element_size() * (world - 1)
        if rank == 0:
            out[k] = torch.cat(lst, dim=0).cpu().numpy()
    if rank != 0:
        return None, moved
    return raw_from_packed(out["headers"], out["root_n"], out["summary"]), moved


def gather_summaries(summary, rank, world):
    """Per-rank emission (BatchedSelfPlayWorker.run, emission "per_rank"): every rank writes the files of its own id range, so
    all that rank 0 needs from the others is what finish_game (self_play.py:219-260) reads - winner, resignation flags, whether
    the game was a no-resign test game: the 32-byte game summaries (engine.GAME_SUMMARY), gathered in rank order = id order;
    the records (304 B per ply) never leave their rank.  Returns (dict with bookkeep_raw's keys, bytes moved) on rank 0,
    (None, bytes) elsewhere."""
    import torch
    import torch.distributed as dist
    from ..engine import GAME_SUMMARY
    dev = summary.device if dist.get_backend() == "nccl" else torch.device("cpu")
    t = summary.contiguous().to(dev)
    lst = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
    dist.gather(t, lst, dst=0)
    moved = t.numel() * t.element_size() * (world - 1)
    if rank != 0:
        return None, moved
    sm = np.ascontiguousarray(torch.cat(lst, dim=0).cpu().numpy()).view(GAME_SUMMARY).reshape(-1)
    return dict(n_plies=sm["n_plies"].copy(), status=sm["status"].copy(), game_id=sm["game_id"].copy(),
                resigned=np.stack([sm["resigned_black"], sm["resigned_white"]], axis=1), enable_resign=sm["enable_resign"].copy()), moved


def wants_fused_tree_net(setting, filters, value_fc, net_reserved, cache_log2, play):
    """Whether the games are stepped by the fused tree + net kernel (csrc/raz_engine_fused.hip).  It applies to 16-filter nets on the
    default net kernels (value_fc_size <= 1024); setting True = wherever it applies, False = never, "auto" = where it applies, the
    evaluation cache (wide nets only anyway) is not attached and the end-game solver is off: with the solver on a game that posts a
    position leaves its launch, and the two-kernel pipeline - whose launches are one step long - gives the solver pool its round
    sooner (mini.yml as shipped, round 5: 25.5 M sims/s two-kernel, 21.3 M fused).  The files are the same either way."""
    if not (filters == 16 and value_fc <= 1024 and net_reserved == 0):
        return False
    if setting is True:
        return True
    solver_on = bool(getattr(play, "use_solver_turn", 0) or getattr(play, "use_solver_turn_in_simulation", 0))
    return setting == "auto" and not cache_log2 and not solver_on


class BatchedSelfPlayWorker:
    """SelfPlayWorker (worker/self_play.py:64-272) for a batch of concurrent games on one GPU."""

    def __init__(self, config, net_blob, games_in_flight=4096, seed=0, device="cuda:0", rank=0, world=1, block_games=None,
                 net_kernel="auto", leaf_cache_log2="auto", leaf_cache_max_discs=24, fused_tree_net="auto", emission="auto"):
        """games_in_flight: game slots resident on the device.  block_games (per rank; default = games_in_flight): the
        number of consecutive game ids a rank plays between two gathers.  With block_games > games_in_flight the slots are
        refilled as games finish (continuous batching, SelfPlayEngine.play_continuous); the files do not depend on either
        number as long as world * block_games - the ids per gather, after which the resign threshold may move - is the same."""
        self.config = config
        self.net_blob = net_blob
        self.games_in_flight = games_in_flight
        self.block_games = block_games or games_in_flight
        # "auto": wide nets (filters % 128 == 0) run their trunk on the f16 matrix cores (raznet-forward-v2, within 1e-5 of
        # the fp32 graph, 3.7x the exact-f32 kernels); "f32": the exact kernels everywhere (engine.DeviceNet)
        self.net_kernel = net_kernel
        # cross-game evaluation cache (include/raz.h raz_engine_set_leaf_cache): "auto" = 2^26 entries (21 GB) for wide nets,
        # whose forward is what a step costs; none for narrow nets (two extra launches per step cost more than they save)
        self.leaf_cache_log2 = leaf_cache_log2
        self.leaf_cache_max_discs = leaf_cache_max_discs
        # 16-filter nets: tree and net in ONE kernel, the game's wave evaluating its own leaves (csrc/raz_engine_fused.hip; the same
        # files bit for bit, +30 % on BASELINE configs[1]).  "auto" = wherever it applies (wants_fused_tree_net); False = the two-kernel pipeline
        self.fused_tree_net = fused_tree_net if fused_tree_net in ("auto", True, False) else bool(fused_tree_net)
        # who writes the files of a block when world > 1 (run()): "rank0" - all records are gathered on rank 0, which writes every
        # file; "per_rank" - every rank writes the files of its own id range and only the 32-byte game summaries are gathered (needs
        # the ids per rank to be a multiple of nb_game_in_file - and of nb_game_in_ggf_file with GGF output - so that no file spans
        # two ranks, and ONE file system under play_data_dir: the ranks of one node); "auto" - per_rank where that holds
        if emission not in ("auto", "rank0", "per_rank"):
            raise ValueError(f"emission={emission!r}: auto, rank0 or per_rank")
        self.emission = emission
        self.seed = seed
        self.device = device
        self.rank, self.world = rank, world
        self.buffer = []
        self.buffer_json = []          # the same rows as JSON text fragments (bytes), one per game (write_raw)
        self._emit_executor = None
        self._emit_executor_threads = 0
        self.move_history_buffer = []
        self.false_positive_count_of_resign = 0
        self.resign_test_game_count = 0
        self._engine = None
        self._net = None
        self._engine_key = None
        # worker/self_play.py:109-111,132-134: with share_mtcs_info_in_self_play the worker keeps one MCTSInfo for
        # reset_mtcs_info_per_game consecutive games (mini.yml: 3).  Here a "worker" is a game slot: the games a
        # slot plays in consecutive play_batch calls form such a series on the slot's device tree.
        self._series_pos = 0
        # The reference steps the threshold after every 100 no-resign test games of ONE process playing one game at a
        # time (self_play.py:250-260).  A batch finishes thousands of games that were all played under the SAME
        # threshold: stepping once per 100 of them would multiply the control gain by the batch size, so the emitters
        # defer the check and apply it once per emitted batch (>= 100 test games accumulated, as in the reference).
        self._defer_threshold_update = False
        # raznet-forward-v2 reported an activation outside the f16 range: this net runs on the exact-f32 kernels from now on
        # (until the next set_net_blob)
        self._f32_fallback = False

    # -- engine life cycle -------------------------------------------------------------------
    def _series_length(self):
        p = self.config.play
        if not p.share_mtcs_info_in_self_play:
            return 1   # mtcs_info stays None: every game's players get fresh trees (self_play.py:109, player.py:43)
        r = getattr(p, "reset_mtcs_info_per_game", 1)
        if not r or int(r) < 1:
            raise ValueError("play.reset_mtcs_info_per_game must be >= 1: 0/None never resets the shared tree "
                             "(worker/self_play.py:132), which a fixed device node pool cannot hold")
        return int(r)

    def _get_engine(self, max_sims):
        from ..engine import DeviceNet, SelfPlayEngine
        # (resign_threshold is a run-time parameter of the engine - raz_engine_set_resign_threshold - not part of
        #  its identity: the reference keeps mtcs_info across threshold updates, worker/self_play.py:250-260)
        key = (max_sims, self.config.play.thinking_loop)
        if self._engine is not None and self._engine_key != key:
            self._drop_engine()   # BEFORE the new pools are sized: the old workspace must be back in the free list
        if self._net is None:
            self._net = DeviceNet(self.net_blob, self.device, kernel="f32" if self._f32_fallback else self.net_kernel)
        if self._engine is None:
            self._series_pos = 0   # a new engine starts from empty trees
            r = self._series_length()
            p = self.config.play
            whole = (max_sims * max(1, p.thinking_loop) * 62 + 128) * 2   # a whole game's tree (two nodes per simulation at most)
            cache = self.leaf_cache_log2
            if cache == "auto":
                cache = 26 if self._net.filters >= 128 else None
            nodes = None
            if r > 1:   # the tree of a series is never pruned (its early positions are searched again): room for r games
                nodes = r * whole
            else:
                # whole-game trees for 8192 games x 800 sims/move would still be ~190 GB of compact nodes + 100 GB of tables:
                # size the pools for the HBM that is there (k_gc prunes the positions the real game has left behind;
                # pools of ~16 x sims nodes suffice)
                cap = self._pool_nodes_that_fit(cache_log2=cache)
                if cap is not None and cap < whole:
                    if cap < 12 * max_sims:
                        raise RuntimeError(f"{self.games_in_flight} games in flight at {max_sims} sims/move leave {cap} tree nodes per "
                                           f"game in this GPU's free memory (< 12 x sims): lower games_in_flight")
                    nodes = cap
            from ..engine import NODE_MAX_BYTES, WHOLE_GAME_BYTES_PER_NODE
            pool_bytes = 0   # pruned pools: the default budget (232 B per node); k_gc is triggered by bytes as well as by count
            if r > 1:        # never pruned: room for whole games whatever their mobility
                want = nodes * WHOLE_GAME_BYTES_PER_NODE + 64 * NODE_MAX_BYTES
                pool_bytes = min(want, 255 << 20)
                if pool_bytes < want:   # the link range caps a pool at 256 MB: size the node count (directory, table) for the bytes that exist
                    nodes = (pool_bytes - 64 * NODE_MAX_BYTES) // WHOLE_GAME_BYTES_PER_NODE
                    logger.warning(f"series of {r} games at {max_sims} sims/move x thinking_loop {p.thinking_loop}: the never-pruned node pool is "
                                   f"capped at 255 MB per game ({nodes} nodes instead of {r * whole}); a series that outgrows it fails with 'pool full'")
                free = self._free_bytes()
                if free is not None and self.games_in_flight * (pool_bytes + nodes * (4 * 32 + 8)) > free:   # pools + table slots + directory
                    raise RuntimeError(f"{self.games_in_flight} games in flight with never-pruned trees of {r} games at {max_sims} sims/move need "
                                       f"{self.games_in_flight * pool_bytes / 2**30:.0f} GiB of node pools; {free / 2**30:.0f} GiB are free: lower games_in_flight")
            # 16-filter nets: tree and net in one kernel unless told otherwise ("auto"; the evaluation cache is for wide nets only)
            fused = wants_fused_tree_net(self.fused_tree_net, self._net.filters, self._net.value_fc, self._net.c.reserved, cache, self.config.play)
            self._engine = SelfPlayEngine(self.config, self._net, self.games_in_flight, seed=self.seed,
                                          sims_hint=max_sims, nodes_per_game=nodes, leaf_cache_log2=None if fused else cache,
                                          leaf_cache_max_discs=self.leaf_cache_max_discs, pool_bytes_per_game=pool_bytes, fused=fused)
            self._engine_key = key
        self._engine.set_resign_threshold(self.config.play.resign_threshold)
        return self._engine

    def _drop_engine(self, net_too=False):
        """Give the engine's workspace (node pools, tables, evaluation cache, net scratch) back to the device: drop every
        reference, collect, and empty torch's caching allocator - torch.cuda.mem_get_info, which sizes the next engine's
        pools, only sees memory the allocator has returned to the driver."""
        import gc
        self._engine = None
        self._engine_key = None
        self._series_pos = 0
        if net_too:
            self._net = None
        gc.collect()
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.empty_cache()
        except Exception:
            pass

    def _fixed_device_bytes(self, cache_log2):
        """What the engine allocates beside the node pools, tables and prune maps: the evaluation cache, the net's scratch
        and the id-ordered outbox of a block (include/raz.h: raz_leaf_cache_bytes, raz_net_scratch_bytes; outbox rows of
        max_plies x (48 + 256) B + a 32-byte summary + a done flag)."""
        from .._native import lib
        rows = self.games_in_flight * max(1, int(getattr(self.config.play, "parallel_search_num", 1) or 1))
        total = 0
        if cache_log2:
            total += int(lib.raz_leaf_cache_bytes(int(cache_log2), rows)) + 256
        if self._net is not None:
            total += int(lib.raz_net_scratch_bytes(self._net.filters, self._net.value_fc, rows))
        if self.block_games > self.games_in_flight:
            total += self.block_games * (72 * (48 + 256) + 33)
        return total

    def _free_bytes(self):
        """Free device memory, or None when torch cannot tell (CPU-side tests with a stub engine)."""
        try:
            import torch
            return torch.cuda.mem_get_info(torch.device(self.device))[0]
        except Exception:
            return None

    def _pool_nodes_that_fit(self, fraction=0.7, cache_log2=None):
        """Tree nodes per game that fit in `fraction` of the device's free memory once the evaluation cache, the net scratch
        and the outbox are taken out.  Per node of nodes_per_game: 232 B of pool (compact nodes: 40 B + 20 B per legal move,
        ~212 B on average) + up to 4 table slots of 32 B (table_slots is the next power of two >= 2 x nodes) + a directory
        and a prune-map word; per game 64 x 704 B of pool slack (include/raz.h raz_engine_workspace_bytes is the exact
        figure).  None when torch cannot tell."""
        try:
            import torch
            free, _ = torch.cuda.mem_get_info(torch.device(self.device))
        except Exception:
            return None
        from ..engine import NODE_MAX_BYTES, NODE_POOL_BYTES_PER_NODE
        per_node = NODE_POOL_BYTES_PER_NODE + 4 * 32 + 4 + 4
        budget = int(free * fraction) - self._fixed_device_bytes(cache_log2) - self.games_in_flight * 64 * NODE_MAX_BYTES
        return max(0, budget) // (self.games_in_flight * per_node)

    def play_batch(self, first_game_idx, n_games=None):
        """Play global game ids first_game_idx + rank*B .. on this rank.  Returns [(plies, summary)]."""
        n = self.games_in_flight if n_games is None else n_games
        base = first_game_idx + self.rank * self.games_in_flight
        sims = np.array([decide_simulation_num_per_move(self.config, base + i) for i in range(self.games_in_flight)],
                        dtype=np.uint32)
        eng = self._get_engine(int(sims.max()))
        r = self._series_length()
        if self._series_pos == 0:
            eng.start(base, sims, n_active=n)
        else:
            eng.next_game(base, sims, n_active=n)   # the slots' trees are kept (reset_mtcs_info_per_game > 1)
        stats = eng.run(allow_gc=(r == 1))
        self._series_pos = (self._series_pos + 1) % r
        recs = eng.records(save_policy_of_tau_1=self.config.play_data.save_policy_of_tau_1)
        self.last_stats = stats
        return recs

    def play_batch_raw(self, first_game_idx, n_games=None, device_records=False):
        """play_batch without the per-ply Python objects: the engine's record arrays (SelfPlayEngine.read_raw),
        cut to the n games played.  device_records: leave the records in HBM and return (engine, n) - the caller
        packs / gathers them there (run())."""
        n = self.games_in_flight if n_games is None else n_games
        base = first_game_idx + self.rank * self.games_in_flight
        sims = np.array([decide_simulation_num_per_move(self.config, base + i) for i in range(self.games_in_flight)],
                        dtype=np.uint32)
        eng = self._get_engine(int(sims.max()))
        r = self._series_length()
        if self._series_pos == 0:
            eng.start(base, sims, n_active=n)
        else:
            eng.next_game(base, sims, n_active=n)
        self.last_stats = eng.run(allow_gc=(r == 1))
        self._series_pos = (self._series_pos + 1) % r
        if device_records:
            return eng, n
        raw = eng.read_raw()
        return {k: raw[k][:n] for k in RAW_KEYS}

    def play_block_continuous(self, first_game_idx):
        """This rank's block of game ids [first + rank * block_games, + block_games) with continuous batching: the slots are
        refilled with the next id as games finish.  Returns the id-ordered device outbox as a `plies -> packed tensors`
        callable for gather_packed (cut to the block's longest game, or to the extent the ranks agreed on)."""
        blk = self.block_games
        base = first_game_idx + self.rank * blk
        sims = simulation_nums_of_ids(self.config, base, blk)
        eng = self._get_engine(int(sims.max()))
        # steps between two polls of the device (statistics + harvest, a host synchronisation each): a step of a 16-filter net is ~40 us,
        # so 64 steps between polls left the host in the loop a sixth of the time (bench worker_end_to_end_config1: the worker's engine at
        # 19.2 M games/h where the same engine reaches 25 M polled every 200 steps); a wide net's step is ~25 ms and 64 is a poll every 1.6 s
        chunk = 64 if (self._net is not None and getattr(self._net, "filters", 256) >= 128) else 256
        # ... but pools are only pruned at polls: a pool that was capped by the HBM budget (as small as 12 x sims nodes) must not be able
        # to fill up inside one chunk - keep what a chunk can add (nodes_per_step per step) under a quarter of the pool
        pool, per_step = getattr(getattr(eng, "cfg", None), "nodes_per_game", 0), getattr(eng, "nodes_per_step", 0)
        if pool and per_step:
            chunk = max(16, min(chunk, int(pool) // (4 * int(per_step))))
        # a single-rank worker ships the block's finished PREFIX to the file writer while the block is still played (streamed emission,
        # run()): rows [0, p) of the id-ordered outbox are done, p advancing with the polls; pieces are whole files (no file is cut)
        sink, self._streamed_rows = getattr(self, "_stream_sink", None), 0
        unit = whole_files_block(self.config, 1)
        piece = max(unit, (int(getattr(self, "stream_piece_games", 0) or max(1024, self.games_in_flight)) // unit) * unit)

        def ship(steps, done_games, st, ob):
            import torch
            from ..engine import raw_from_packed
            at = self._streamed_rows
            if done_games - at < piece:
                return
            open_rows = (ob["done"][at:] == 0).nonzero()
            p = at + (int(open_rows[0].item()) if open_rows.numel() else blk - at)
            n = ((p - at) // piece) * piece
            if n <= 0:
                return
            plies = max(1, int(ob["summary"][at:at + n, 20:24].contiguous().view(torch.int32).max().item()))
            sink(raw_from_packed(*(t.cpu().numpy() for t in (ob["headers"][at:at + n, :plies].contiguous(), ob["root_n"][at:at + n, :plies].contiguous(),
                                                             ob["summary"][at:at + n]))), at)
            self._streamed_rows = at + n
        outbox, self.last_stats = eng.play_continuous(base, blk, sims, chunk=chunk, on_chunk=ship if sink is not None else None)
        # per block: what the engine's loop took (bench.py worker_end_to_end_config1 reports the run's blocks, not only the last one)
        self.block_stats = getattr(self, "block_stats", [])
        st = self.last_stats
        self.block_stats.append({"games": int(st.get("finished_games", blk)), "sims": int(st.get("total_sims", 0)), "steps": int(st.get("steps", 0)),
                                 **{k: float(v) for k, v in (st.get("seconds") or {}).items()}})
        del self.block_stats[:-64]   # (a worker runs for days)

        def packed(plies, first_row=0):
            """Rows [first_row, block) of the outbox, cut to `plies` (None: their longest game)."""
            if plies is None:   # n_plies is the u32 at byte 20 of a raz_game_summary
                import torch
                plies = max(1, int(outbox["summary"][first_row:, 20:24].contiguous().view(torch.int32).max().item())) if first_row < blk else 1
            return {"headers": outbox["headers"][first_row:, :plies].contiguous(), "root_n": outbox["root_n"][first_row:, :plies].contiguous(),
                    "summary": outbox["summary"][first_row:]}
        return packed

    def emit_raw(self, raw, first_local_idx=1, threads=None):
        """emit() on raw record arrays: the same files, with the rows' JSON text produced natively
        (csrc/raz_emit.hip; one call per game, spread over host threads - ctypes releases the GIL).
        = bookkeep_raw (resignation accounting, one threshold step per batch) + write_raw (the files)."""
        self.bookkeep_raw(raw)
        return self.write_raw(raw, first_local_idx, threads)

    def bookkeep_raw(self, raw):
        """finish_game (self_play.py:219-260) for every game of a batch, then ONE threshold step: all that the next
        batch needs from this one.  Cheap (no rows are touched), so run() does it before handing the batch's arrays to
        the background writer."""
        self._defer_threshold_update = True
        for st, rb, rw, er in zip(raw["status"].tolist(), raw["resigned"][:, 0].tolist(), raw["resigned"][:, 1].tolist(),
                                  raw["enable_resign"].tolist()):
            self.finish_game({"winner": int(st) & 0x0f, "resigned_black": int(rb), "resigned_white": int(rw), "enable_resign": int(er)})
        self._defer_threshold_update = False
        self.check_and_update_resignation_threshold()

    def _emit_pool(self, threads=None):
        import concurrent.futures as cf
        want = threads or min(32, os.cpu_count() or 1)
        if self._emit_executor is None or self._emit_executor_threads != want:
            if self._emit_executor is not None:
                self._emit_executor.shutdown(wait=True)
            self._emit_executor, self._emit_executor_threads = cf.ThreadPoolExecutor(max_workers=want), want
        return self._emit_executor

    def write_raw(self, raw, first_local_idx=1, threads=None, ahead=128, stamp_base=None):
        """The files of a batch of finished games (self_play.py:180-207), streamed: the JSON text of at most `ahead`
        games is in flight on the host threads while the games before them are written, so memory stays bounded by
        nb_game_in_file + ahead games (a game's rows are ~0.7 MB of text) whatever the batch size.
        stamp_base (per-rank emission): the k-th play file and the k-th GGF file of this call are named stamp_base + k
        microseconds instead of by this process's clock - rank 0 hands every rank of a block its own range of stamps, so the
        directory listing sorts in game order whichever rank wrote what when."""
        from collections import deque
        pd, pc, rc = self.config.play_data, self.config.play, self.config.resource
        n = len(raw["n_plies"])
        winners = [int(v) & 0x0f for v in raw["status"].tolist()]
        game_ids = raw["game_id"].tolist()
        ex = self._emit_pool(threads)

        def frag(g):
            return game_rows_json(raw["headers"][g], raw["root_n"][g], raw["n_plies"][g], winners[g],
                                  pc.change_tau_turn, pd.save_policy_of_tau_1, as_bytes=True)
        futs, submitted = deque(), 0
        paths = []
        known_files = None   # the directory is listed once per batch, then tracked (the reference lists it after every game)
        # a finished file is WRITTEN on the host threads too (its text is 0.7 MB per game: one thread writing 64-game files kept the
        # whole worker at 1.3 GB/s on the mini net); names are stamped here, in game order, so the listing sorts as the reference's does
        writes, max_writes = deque(), 8
        stamped = [0, 0]   # play files, GGF files named from stamp_base so far

        def ggf_stamp():
            if stamp_base is None:
                return None
            stamped[1] += 1
            return stamp_base + timedelta(microseconds=stamped[1] - 1)

        def put(path, frags):
            with open(path, "wb") as f:
                f.write(b"[")
                for i, t in enumerate(frags):
                    if i:
                        f.write(b", ")
                    f.write(t)
                f.write(b"]")
                return f.tell()

        def settle(leave):
            while len(writes) > leave:
                self.bytes_written = getattr(self, "bytes_written", 0) + writes.popleft().result()
        for g in range(n):
            while submitted < min(n, g + ahead):
                futs.append(ex.submit(frag, submitted))
                submitted += 1
            text, nrows = futs.popleft().result()
            local_idx = first_local_idx + g
            is_draw = nrows > 0 and winners[g] == 3          # rows[0][-1] == 0 (self_play.py:181)
            if nrows and (not is_draw or pd.drop_draw_game_rate <= drop_draw_uniform(self.seed, int(game_ids[g]))):
                self.buffer_json.append(text)
            if local_idx % pd.nb_game_in_file == 0 and self.buffer_json:
                last = getattr(self, "_last_file_stamp", None)
                if stamp_base is not None:
                    now = stamp_base + timedelta(microseconds=stamped[0])
                    stamped[0] += 1
                else:
                    now = datetime.now()
                    if last is not None and now <= last:   # (two files inside one clock tick must still sort in game order)
                        now = last + timedelta(microseconds=1)
                self._last_file_stamp = now if last is None else max(now, last)
                path = os.path.join(rc.play_data_dir, rc.play_data_filename_tmpl % now.strftime("%Y%m%d-%H%M%S.%f"))
                writes.append(ex.submit(put, path, self.buffer_json))
                self.buffer_json = []
                paths.append(path)
                if known_files is None:
                    settle(0)   # (the listing must see this batch's first file)
                    known_files = get_game_data_filenames(rc)
                elif not known_files or known_files[-1] < path:
                    known_files.append(path)
                elif path not in known_files:   # (per-rank emission: files another rank wrote ahead of this one carry later stamps)
                    import bisect
                    bisect.insort(known_files, path)
                settle(0 if pd.max_file_num <= max_writes else max_writes)   # never prune a file that is still being written
                known_files = self.remove_play_data(known_files)
            if pd.enable_ggf_data:
                wr = (local_idx % pd.nb_game_in_ggf_file == 0) or local_idx <= 5
                self.save_ggf_data(plies_for_ggf(raw["headers"][g], raw["n_plies"][g]), write=wr, stamp=ggf_stamp() if wr else None)
        settle(0)
        return paths

    # -- bookkeeping identical to the reference worker ---------------------------------------------
    def finish_game(self, summary):
        """self_play.py:219-260: resign false-positive accounting over the no-resign test games."""
        w = summary["winner"]
        if w == 1:
            fp = bool(summary["resigned_black"])
        elif w == 2:
            fp = bool(summary["resigned_white"])
        else:
            fp = bool(summary["resigned_black"] or summary["resigned_white"])
        if not summary["enable_resign"]:
            self.resign_test_game_count += 1
            if fp:
                self.false_positive_count_of_resign += 1
            if not self._defer_threshold_update:
                self.check_and_update_resignation_threshold()

    def check_and_update_resignation_threshold(self):
        pc = self.config.play
        if self.resign_test_game_count < 100 or pc.resign_threshold is None:
            return
        rate = self.false_positive_count_of_resign / self.resign_test_game_count
        if rate >= pc.false_positive_threshold:
            pc.resign_threshold -= pc.resign_threshold_delta
        else:
            pc.resign_threshold += pc.resign_threshold_delta
        self.false_positive_count_of_resign = 0
        self.resign_test_game_count = 0

    def save_play_data(self, plies, summary, write=True):
        """self_play.py:180-194."""
        rows = rows_of_game(plies, summary["winner"])
        is_draw = bool(rows) and rows[0][-1] == 0
        if not is_draw or self.config.play_data.drop_draw_game_rate <= drop_draw_uniform(self.seed, summary["game_id"]):
            self.buffer += rows
        if not write or not self.buffer:
            return None
        rc = self.config.resource
        game_id = datetime.now().strftime("%Y%m%d-%H%M%S.%f")
        path = os.path.join(rc.play_data_dir, rc.play_data_filename_tmpl % game_id)
        write_game_data_to_file(path, self.buffer)
        self.buffer = []
        return path

    def save_ggf_data(self, plies, write=True, stamp=None):
        """self_play.py:196-207.  stamp: the file's name (write_raw's stamp_base) instead of this process's clock."""
        self.move_history_buffer.append(ggf_moves_of_game(plies))
        if not write:
            return None
        rc = self.config.resource
        game_id = (stamp or datetime.now()).strftime("%Y%m%d-%H%M%S.%f")
        path = os.path.join(rc.self_play_ggf_data_dir, rc.ggf_filename_tmpl % game_id)
        with open(path, "wt") as f:
            for moves in self.move_history_buffer:
                f.write(make_ggf_string("RAZ", "RAZ", moves=moves) + "\n")
        self.move_history_buffer = []
        return path

    def remove_play_data(self, files=None):
        """self_play.py:209-217.  `files`: the sorted listing if the caller tracks it; returns what is left."""
        if files is None:
            files = get_game_data_filenames(self.config.resource)
        extra = len(files) - self.config.play_data.max_file_num
        if extra <= 0:
            return files
        for i in range(extra):
            try:
                os.remove(files[i])
            except OSError:
                pass
        return files[extra:]

    def emit(self, records, first_local_idx=1):
        """Write one batch of finished games the way the reference loop would, game by game."""
        pd = self.config.play_data
        paths = []
        self._defer_threshold_update = True
        for k, (plies, summary) in enumerate(records):
            local_idx = first_local_idx + k
            self.finish_game(summary)
            p = self.save_play_data(plies, summary, write=local_idx % pd.nb_game_in_file == 0)
            if p:
                paths.append(p)
            self.remove_play_data()
            if pd.enable_ggf_data:
                self.save_ggf_data(plies, write=(local_idx % pd.nb_game_in_ggf_file == 0) or local_idx <= 5)
        self._defer_threshold_update = False
        self.check_and_update_resignation_threshold()
        return paths

    def _play_block(self, game_idx):
        """This rank's ids of the block that starts at game_idx.  Returns (packed, ids per rank): `packed` is the
        `plies -> packed device tensors` callable gather_packed wants."""
        continuous = self.block_games > self.games_in_flight and self._series_length() == 1
        if continuous:
            return self.play_block_continuous(game_idx), self.block_games
        eng, n = self.play_batch_raw(game_idx, device_records=True)
        return (lambda plies, eng=eng, n=n: eng.pack_records(0, n, plies)), self.games_in_flight

    BLOCK_OK, BLOCK_RANGE, BLOCK_FATAL = 0, 1, 2

    def _play_block_checked(self, game_idx):
        """_play_block + what became of it on THIS rank: BLOCK_OK; BLOCK_RANGE - the net's range flag is up (a search fed with
        out-of-range priors may also trip an engine error flag before the block ends: the same event; the block is void either
        way and is played again on the exact-f32 kernels); BLOCK_FATAL - anything else went wrong (pool / table / records full with
        the net in range, a HIP error, ...): the exception is kept in self._block_error and raised once every rank knows, so
        that no rank is left waiting in a collective for one that has gone."""
        self._block_error = None
        try:
            packed, per_rank = self._play_block(game_idx)
            return packed, per_rank, (self.BLOCK_OK if self._net.range_ok() else self.BLOCK_RANGE)
        except Exception as ex:   # noqa: BLE001 - re-raised by _all_ranks_state on every rank
            in_range = True
            try:
                in_range = self._net is None or bool(self._net.range_ok())
            except Exception:   # noqa: BLE001 - the device may be gone: fatal
                pass
            if isinstance(ex, RuntimeError) and not in_range:
                return None, 0, self.BLOCK_RANGE
            self._block_error = ex
            return None, 0, self.BLOCK_FATAL

    def _group(self):
        """True when the collectives of the path run: more than one rank, or a process group of one (the nccl path exercised on a
        single GPU - tests/test_multirank_gpu.py, bench.py RAZ_BENCH_NCCL_WORLD1).  The process group must be THIS worker's group:
        a host process that initialised torch.distributed for more ranks than the worker was built for would otherwise walk a
        world-1 worker into collectives of a larger group (size mismatch or hang)."""
        try:
            import torch.distributed as dist
            up = dist.is_available() and dist.is_initialized()
        except Exception:   # noqa: BLE001
            up = False
        if up and dist.get_world_size() != self.world:
            raise RuntimeError(f"BatchedSelfPlayWorker(world={self.world}) inside a torch.distributed process group of {dist.get_world_size()} "
                               "ranks: build the worker with that world size (rank = its rank) or destroy the group first")
        return self.world > 1 or up

    def _all_ranks_state(self, state, blocks_written=0):
        """MAX of the per-rank block state over all ranks: every rank takes the same branch before the next collective, and a
        failure anywhere raises everywhere (the failing rank its own exception, the others a RuntimeError naming the event).
        The same collective carries the MIN over ranks of `blocks_written` (per-rank emission: how many blocks' files this rank
        has on disk) into self._blocks_written_everywhere - what rank 0 may advance data/.self-play-game-idx to."""
        worst = state
        if not self._group():
            self._blocks_written_everywhere = blocks_written
        else:   # (set only from the reduced value: a collective that fails leaves the last agreed count)
            import torch
            import torch.distributed as dist
            dev = torch.device(self.device) if dist.get_backend() == "nccl" else torch.device("cpu")
            t = torch.tensor([state, -int(blocks_written)], dtype=torch.int32, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            worst, least = (int(v) for v in t.tolist())
            self._blocks_written_everywhere = -least
        if worst == self.BLOCK_FATAL:
            if self._block_error is not None:
                raise self._block_error
            raise RuntimeError(f"rank {self.rank}: another rank failed while playing its block; this rank stops with it")
        return worst

    def _broadcast_blob(self, blob):
        """Rank 0's decision about a new generation of weights, and the weights themselves, to every rank: ranks polling
        the model files on their own could see the optimizer's half-written file, or see it at different times, and play
        the same block with different nets."""
        import torch
        import torch.distributed as dist
        meta = [len(blob) if blob is not None else 0]
        dist.broadcast_object_list(meta, src=0)
        if not meta[0]:
            return None
        dev = torch.device(self.device) if dist.get_backend() == "nccl" else torch.device("cpu")
        if self.rank == 0:
            t = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
        else:
            t = torch.empty(meta[0], dtype=torch.uint8, device=dev)
        dist.broadcast(t, src=0)
        return blob if self.rank == 0 else t.cpu().numpy().tobytes()

    def _may_replay_block(self):
        """True when a block may be discarded and played again - the net's trunk runs on the split-f16 kernels, whose range flag is
        looked at after the block (run()): nothing of such a block may reach the disk before it is complete."""
        if self._f32_fallback or self.net_kernel == "f32":
            return False
        import struct
        filters = struct.unpack_from("<5i", self.net_blob, 0)[2] if self.net_blob and len(self.net_blob) >= 20 else 256
        return self.net_kernel in ("auto", "f16x3") and filters >= 128 and filters % 128 == 0

    def _ids_per_block(self):
        """Consecutive game ids ONE rank plays between two gathers (_play_block)."""
        continuous = self.block_games > self.games_in_flight and self._series_length() == 1
        return self.block_games if continuous else self.games_in_flight

    def _per_rank_emission(self):
        """Whether every rank writes the files of its own id range (see __init__, `emission`).  A file holds nb_game_in_file
        consecutive games counted from the run's first (self_play.py:186: local index % nb_game_in_file), so a rank's range is a
        whole number of files exactly when its length is a multiple of that number; the GGF buffer likewise."""
        if self.world <= 1 or self.emission == "rank0":
            return False
        if self.emission == "auto" and not self._ranks_share_one_host():
            # ranks on several nodes: each would write into its OWN node's play_data_dir while rank 0's game index and pruning count
            # every file as local, and the optimizer on node 0 would train on 1 / nodes of the games.  Gather everything on
            # rank 0 instead; emission="per_rank" stays available to a caller who knows play_data_dir is one shared file system
            if not getattr(self, "_said_rank0", False):
                self._said_rank0 = True
                logger.info("emission='auto': the ranks span several hosts (LOCAL_WORLD_SIZE < WORLD_SIZE) - every record is gathered on rank 0, which writes all "
                            "files; pass emission='per_rank' if play_data_dir is one file system shared by all nodes")
            return False
        pd, n = self.config.play_data, self._ids_per_block()
        aligned = n % pd.nb_game_in_file == 0 and (not pd.enable_ggf_data or n % pd.nb_game_in_ggf_file == 0)
        if self.emission == "per_rank" and not aligned:
            raise ValueError(f"emission='per_rank': {n} game ids per rank and block are not a multiple of nb_game_in_file {pd.nb_game_in_file}"
                             + (f" and nb_game_in_ggf_file {pd.nb_game_in_ggf_file}" if pd.enable_ggf_data else "") + " - a file would span two ranks")
        return aligned

    def _ranks_share_one_host(self):
        """True when all ranks of the job run on this host - the launcher's LOCAL_WORLD_SIZE equals the world size (torchrun sets both;
        without a launcher's environment the ranks were started by hand on one machine: tests, bench rigs)."""
        lws = os.environ.get("LOCAL_WORLD_SIZE")
        return lws is None or int(lws) >= self.world

    def _next_block_stamps(self, per_rank):
        """Rank 0, per-rank emission: the first file-name stamp of a block; rank r names its files from stamp + r * (per_rank + 8)
        microseconds on (write_raw's stamp_base; at most per_rank / nb_game_in_ggf_file + 5 files per rank), so names sort by game
        id across ranks and blocks, and after every file an earlier block or run left."""
        now = datetime.now()
        for last in (getattr(self, "_block_stamps_end", None), getattr(self, "_last_file_stamp", None)):
            if last is not None and now <= last:
                now = last + timedelta(microseconds=1)
        self._block_stamps_end = now + timedelta(microseconds=self.world * (per_rank + 8))
        return now

    def run(self, total_games=None, reload_model=None, background_emit=True, until=None):
        """_start (self_play.py:95-137): play batches until total_games (None = forever) - or, with `until` (a time.monotonic()
        deadline; rank 0's clock decides for all ranks), until the block during which the deadline passes has been written.
        Per batch: every rank plays its id range; the finished games' records are packed in HBM and gathered on
        rank 0 (the path's single collective), which does the resignation bookkeeping, broadcasts what every rank needs
        for the next batch - the game index and the resign threshold (so that all ranks keep playing under the same
        rule and results stay independent of the sharding) - and writes the files.  With background_emit the files of
        batch k are written by a host thread while batch k + 1 is played (the text of 65 536 games is ~45 GB: minutes
        of host work that would otherwise idle every GPU); data/.self-play-game-idx is advanced after a batch's files
        are on disk, and run() returns only when everything is written.
        Per-rank emission (`emission`, _per_rank_emission): the records stay where they were played - each rank copies its own to
        its host and writes the files of its own id range (its own PCIe link, its own host threads; rank 0 would otherwise turn
        world x 45 GB of text per block on its own) - and the gather shrinks to the 32-byte game summaries rank 0's bookkeeping
        reads; rank 0 adds the block's file-name stamps to the broadcast, advances data/.self-play-game-idx to a block once the
        block-state all_reduce has shown every rank's files of it on disk, and prunes to max_file_num once more at the end (each
        rank prunes from its own listing meanwhile, which can only leave too many files, never remove a wrong one).  The files
        are those of the rank-0 path byte for byte (tests/test_worker_run_host.py).
        reload_model (optional): callable returning a new net blob or None, polled by RANK 0 between batches
        (api.py:117-125); its answer - and the weights - are broadcast, so every rank plays a block with the same net.
        A block during which an activation of the split-f16 trunk left the f16 range (raz_net_range_check) is discarded
        on every rank and played again on the exact-f32 kernels, which the net then stays on (include/raz.h)."""
        import time as _tm
        import torch.distributed as dist
        from ..engine import raw_from_packed
        rc = self.config.resource
        if self.rank == 0:
            rc.create_directories()
        game_idx = read_as_int(rc.self_play_game_idx_file) or 0
        local_idx = 1
        own_files = self._per_rank_emission()
        # Streamed emission (one rank, background writer, continuous blocks, a net on the exact-f32 kernels - a block of the split-f16
        # trunk may still be discarded and replayed, below): the finished prefix of a block goes to the writer in pieces of whole files
        # WHILE the block is played, so the writer works evenly through the block and what is left to write when the last game ends is
        # one piece, not the block (65 536 games of a 16-filter net are 14 GB of text: 7 s of a 9 s block behind its end).
        stream = (background_emit and self.world == 1 and not own_files and self._ids_per_block() > self.games_in_flight
                  and not self._may_replay_block())
        # (pieces of <= a quarter of the slots: the play loop must not wait for the writer - 16 may queue up)
        writer = _BackgroundWriter(self, depth=16 if stream else 1) if (background_emit and (self.rank == 0 or own_files)) else None
        inline = {"written": 0, "error": None}   # per-rank emission without the background thread
        block_game_idx = []                      # rank 0, per-rank emission: the game index after each block

        def written():   # blocks whose files THIS rank has on disk
            return writer.batches if writer is not None else inline["written"]

        def with_writer(state):
            """Per-rank emission: a rank whose writer has failed says so in the block state, so that every rank stops with it (raised
            on that rank alone, the others would wait for it in the next collective)."""
            if own_files and state != self.BLOCK_FATAL:
                err = inline["error"] if writer is None else (writer.take_error() if writer.failed else None)
                if err is not None:
                    self._block_error = err
                    return self.BLOCK_FATAL
            return state

        def advance_game_idx():
            done = min(self._blocks_written_everywhere, len(block_game_idx))
            if self.rank == 0 and own_files and done > advance_game_idx.done:
                advance_game_idx.done = done
                self._write_game_idx(block_game_idx[done - 1])
        advance_game_idx.done = 0
        self._blocks_written_everywhere = 0

        def agree(state):
            """The block state all ranks share (raises on every rank if one of them failed) - and, failed or not, the game index
            file advanced to the last block the same collective showed on disk everywhere."""
            try:
                return self._all_ranks_state(with_writer(state), written())
            finally:
                advance_game_idx()
        # The writer thread runs Python between its native calls; the play loop gets the interpreter back only when that thread gives
        # it up - by default every 5 ms, which is half a poll interval of a 16-filter net (bench worker_end_to_end_config1: a block played
        # beside a busy writer took 9.4 s, the first block, with nothing to write yet, 7.5 s).  Hand over every 0.2 ms while run() lasts.
        import sys as _sys
        _switch = _sys.getswitchinterval()
        if writer is not None:
            _sys.setswitchinterval(min(_switch, 2e-4))
        finished = False
        try:
            while total_games is None or local_idx <= total_games:
                if until is not None:
                    stop = [_tm.monotonic() >= until]
                    if self._group():
                        dist.broadcast_object_list(stop, src=0)
                    if stop[0]:
                        break
                _t0 = _tm.monotonic()
                self._stream_sink = (lambda raw, row, _l=local_idx: writer.submit(raw, _l + row, None)) if stream else None
                packed, per_rank, state = self._play_block_checked(game_idx)
                _t1 = _tm.monotonic()
                while agree(state) == self.BLOCK_RANGE:
                    if self._f32_fallback or self._series_length() > 1:
                        raise RuntimeError("the net left its numeric range on the exact-f32 kernels too" if self._f32_fallback else
                                           "an activation of the net left the f16 range of the split-operand trunk (raznet-forward-v2) "
                                           "inside a series of games on a carried tree: run the worker with net_kernel='f32'")
                    logger.warning("an activation left the f16 range of the split-operand trunk (raznet-forward-v2): the block at "
                                   f"game index {game_idx} is played again on the exact-f32 kernels, which this net stays on")
                    del packed
                    self._f32_fallback = True
                    self._drop_engine(net_too=True)
                    packed, per_rank, state = self._play_block_checked(game_idx)
                mine = None
                if own_files:
                    pk = packed(None)   # cut to THIS rank's longest game: nothing has to agree in shape but the summaries
                    mine = raw_from_packed(*(pk[k].cpu().numpy() for k in ("headers", "root_n", "summary")))
                    allraw, self.last_gather_bytes = gather_summaries(pk["summary"], self.rank, self.world)
                    self.last_gather_backend = dist.get_backend()
                    del pk
                elif stream and state == self.BLOCK_OK:
                    # the rest of the block (what was not shipped while it was played) + every game's summary for the bookkeeping
                    shipped = int(getattr(self, "_streamed_rows", 0))
                    pk = packed(None, shipped)
                    rest = raw_from_packed(*(pk[k].cpu().numpy() for k in ("headers", "root_n", "summary")))
                    if self._group():   # (a process group of one rank: the summaries take the collective's path, 32 B per game)
                        allraw, self.last_gather_bytes = gather_summaries(packed(1)["summary"], self.rank, self.world)
                        self.last_gather_backend = dist.get_backend()
                    else:
                        from ..engine import GAME_SUMMARY
                        sm = np.ascontiguousarray(packed(1)["summary"].cpu().numpy()).view(GAME_SUMMARY).reshape(-1)
                        allraw = dict(n_plies=sm["n_plies"].copy(), status=sm["status"].copy(), game_id=sm["game_id"].copy(),
                                      resigned=np.stack([sm["resigned_black"], sm["resigned_white"]], axis=1), enable_resign=sm["enable_resign"].copy())
                    del pk
                elif self._group():
                    allraw, self.last_gather_bytes = gather_packed(packed, self.rank, self.world)
                    self.last_gather_backend = dist.get_backend()
                else:
                    pk = packed(None)
                    allraw = raw_from_packed(*(pk[k].cpu().numpy() for k in ("headers", "root_n", "summary")))
                    del pk
                del packed   # (holds the engine / the outbox: nothing of this block may pin device memory past this point)
                _t2 = _tm.monotonic()
                stamps = None
                if self.rank == 0:
                    self.bookkeep_raw(allraw)
                    _t3 = _tm.monotonic()
                    game_idx += len(allraw["n_plies"])
                    if own_files:
                        stamps = self._next_block_stamps(per_rank)
                        block_game_idx.append(game_idx)
                    elif stream:
                        writer.submit(rest, local_idx + shipped, game_idx)   # (the index file moves once the block's last file is written)
                        del rest
                    elif writer is not None:
                        writer.submit(allraw, local_idx, game_idx)
                    else:
                        self.write_raw(allraw, local_idx)
                        self._write_game_idx(game_idx)
                    # where a block's wall time went on this rank (bench.py worker_end_to_end_config1): playing, the gather + the
                    # records' copy to the host, the resignation bookkeeping, waiting for the writer to take the block
                    acc = self.run_seconds = getattr(self, "run_seconds", None) or {"play": 0.0, "gather_and_copy": 0.0, "bookkeeping": 0.0, "handing_to_the_writer": 0.0}
                    for k_, v_ in (("play", _t1 - _t0), ("gather_and_copy", _t2 - _t1), ("bookkeeping", _t3 - _t2), ("handing_to_the_writer", _tm.monotonic() - _t3)):
                        acc[k_] += v_
                if self._group():
                    t = [game_idx, self.config.play.resign_threshold, stamps]
                    dist.broadcast_object_list(t, src=0)
                    game_idx, self.config.play.resign_threshold, stamps = t
                if own_files:
                    first_local = local_idx + self.rank * per_rank
                    base = stamps + timedelta(microseconds=self.rank * (per_rank + 8))
                    if writer is not None:
                        writer.submit(mine, first_local, None, quiet=True, stamp_base=base)
                    elif inline["error"] is None:
                        try:
                            self.write_raw(mine, first_local, stamp_base=base)
                            inline["written"] += 1
                        except Exception as ex:   # noqa: BLE001 - agreed on by all ranks in the next block state
                            inline["error"] = RuntimeError(f"writing play data failed: {ex!r}")
                    del mine
                local_idx += per_rank * self.world
                if reload_model is not None:
                    blob = reload_model() if self.rank == 0 else None   # rank 0 decides; the digest of ITS read is what counts
                    if self._group():
                        blob = self._broadcast_blob(blob)
                    if blob is not None:
                        self.set_net_blob(blob)
            finished = True
        finally:
            _sys.setswitchinterval(_switch)
            if writer is not None:
                try:
                    writer.close(quiet=own_files)
                finally:
                    self.last_writer_busy_seconds, self.last_writer_batches = writer.busy_seconds, writer.batches
        if own_files and finished:
            # every rank's writer is closed: agree that all files are on disk (or fail together), then rank 0 finishes the run
            agree(self.BLOCK_OK)
            if self.rank == 0:
                self.remove_play_data()

    def _write_game_idx(self, game_idx):
        with open(self.config.resource.self_play_game_idx_file, "wt") as f:
            f.write(str(game_idx))

    def set_net_blob(self, blob):
        """A new generation of weights between two batches (agent/api.py:117-125 try_reload_model): the engine is
        rebuilt on the new net at the next batch (a tree searched with the old priors is not carried over).  The old
        engine's device memory is released first, so that the new pools are sized against what is really free."""
        self.net_blob = blob
        self._f32_fallback = False
        self._drop_engine(net_too=True)


class _BackgroundWriter:
    """One host thread that writes finished batches, in submission order, while the next batch is played.  At most one
    batch waits behind the one being written (submit blocks beyond that), so at most three batches' record arrays are
    alive.  An error in the thread is raised in the caller at the next submit() / close().  Once a batch has failed the
    writer is dead for good (`failed` is sticky and owned by the thread; the caller only ever takes the exception out of
    `error`): every batch queued behind the failed one is dropped, so data/.self-play-game-idx can never advance past a
    batch whose files are missing."""

    def __init__(self, worker, depth=1):
        import queue
        import threading
        self.worker, self.error = worker, None
        self.busy_seconds = 0.0        # time the thread spent writing batches (bench: the writer's share of the run)
        self.batches = 0
        self.failed = False            # written by the writer thread only, never cleared
        self._reported = False         # the caller has been handed the exception
        self._lock = threading.Lock()  # guards `error` (handed from the thread to the caller exactly once)
        self.queue = queue.Queue(maxsize=depth)   # (1: whole blocks; more: the pieces of streamed emission, run())
        self.thread = threading.Thread(target=self._loop, name="raz-play-data-writer", daemon=True)
        self.thread.start()

    def _loop(self):
        while True:
            item = self.queue.get()
            if item is None:
                return
            if self.failed:
                continue   # keep draining so that submit() never blocks forever
            raw, local_idx, game_idx, kw = item
            try:
                import time as _time
                t0 = _time.monotonic()
                self.worker.write_raw(raw, local_idx, **kw)
                if game_idx is not None:   # (per-rank emission: rank 0 advances the index once EVERY rank has written the block)
                    self.worker._write_game_idx(game_idx)
                self.busy_seconds += _time.monotonic() - t0
                self.batches += 1
            except BaseException as e:   # noqa: B902 - reported to the caller
                with self._lock:
                    self.error = e
                self.failed = True

    def _check(self, closing=False):
        with self._lock:
            err, self.error = self.error, None
        if err is not None:
            self._reported = True
            raise RuntimeError(f"writing play data failed: {err!r}") from err
        if self.failed and not (closing and self._reported):   # (close() after the error was raised: nothing new to say)
            raise RuntimeError("writing play data failed earlier: the writer is stopped")

    def submit(self, raw, local_idx, game_idx, quiet=False, **write_raw_kw):
        """quiet (ranks of a process group): a failure is not raised here, on this rank alone - the caller folds `failed` into the
        block state every rank agrees on (take_error) - and a batch submitted to a failed writer is dropped like any other."""
        if not quiet:
            self._check()
        self.queue.put((raw, local_idx, game_idx, write_raw_kw))

    def take_error(self):
        """The thread's exception as the RuntimeError _check would raise (None if the writer is healthy), without raising."""
        try:
            self._check()
        except RuntimeError as ex:
            return ex
        return None

    def close(self, quiet=False):
        self.queue.put(None)
        self.thread.join()
        if not quiet:
            self._check(closing=True)


# ---- the single collective: finished-game records -> rank 0 ------------------------------------------
def pack_records(records):
    """Fixed-layout numpy packing of [(plies, summary)] for the gather: per game a 64-byte summary and
    per ply 48 B header + 64 x u32 visit counts + 64 x f64 saved policy."""
    n = len(records)
    maxp = max((len(p) for p, _ in records), default=0)
    summ = np.zeros((n, 8), dtype=np.int64)
    hdr = np.zeros((n, maxp, 6), dtype=np.float64)   # player turn action has_row n q  (+ own/enemy below)
    boards = np.zeros((n, maxp, 2), dtype=np.uint64)
    sims = np.zeros((n, maxp, 3), dtype=np.int64)
    rootn = np.zeros((n, maxp, 64), dtype=np.float64)
    pol = np.zeros((n, maxp, 64), dtype=np.float64)
    for i, (plies, s) in enumerate(records):
        summ[i] = [s["winner"], s["status"], len(plies), s["game_id"], s["enable_resign"], s["resigned_black"],
                   s["resigned_white"], 0]
        boards_final = (s["black"], s["white"])
        for j, p in enumerate(plies):
            hdr[i, j] = [p["player"], p["turn"], p["action"], int(p["has_row"]), p["n"], p["q"]]
            boards[i, j] = [p["own"], p["enemy"]]
            sims[i, j] = [p["sims"], p["loops"], int(bool(p.get("solved", False)))]
            rootn[i, j] = p["root_n"]
            pol[i, j] = p["saved_policy"]
        summ[i, 7] = 0
    finals = np.array([[s["black"], s["white"]] for _, s in records], dtype=np.uint64).reshape(n, 2)
    return {"summ": summ, "hdr": hdr, "boards": boards, "sims": sims, "rootn": rootn, "pol": pol, "finals": finals}


def unpack_records(pk):
    out = []
    for i in range(pk["summ"].shape[0]):
        w, status, npl, gid, er, rb, rw, _ = (int(v) for v in pk["summ"][i])
        plies = []
        for j in range(npl):
            pl, turn, act, hr, n, q = pk["hdr"][i, j]
            plies.append({"player": int(pl), "turn": int(turn), "own": int(pk["boards"][i, j, 0]),
                          "enemy": int(pk["boards"][i, j, 1]), "action": int(act), "has_row": bool(hr),
                          "sims": int(pk["sims"][i, j, 0]), "loops": int(pk["sims"][i, j, 1]),
                          "solved": bool(pk["sims"][i, j, 2]), "n": float(n),
                          "q": float(q), "root_n": [float(v) for v in pk["rootn"][i, j]], "root_w": None,
                          "saved_policy": [float(v) for v in pk["pol"][i, j]]})
        out.append((plies, {"winner": w, "status": status, "plies": npl, "game_id": gid, "enable_resign": er,
                            "resigned_black": rb, "resigned_white": rw, "black": int(pk["finals"][i, 0]),
                            "white": int(pk["finals"][i, 1])}))
    return out


def gather_records(records, rank, world, device=None):
    """Gather every rank's finished-game records on rank 0 (returns [] elsewhere), ordered by rank,
    i.e. by global game id.  One all_gather of sizes + one gather per packed array — the only
    communication in the whole self-play path."""
    import torch
    import torch.distributed as dist
    pk = pack_records(records)
    backend = dist.get_backend()
    dev = torch.device(device) if device is not None else (
        torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu"))
    shape = torch.tensor([pk["summ"].shape[0], pk["hdr"].shape[1]], dtype=torch.int64, device=dev)
    shapes = [torch.zeros_like(shape) for _ in range(world)]
    dist.all_gather(shapes, shape)
    max_n = int(max(s[0] for s in shapes))
    max_p = int(max(s[1] for s in shapes))
    gathered = {}
    for k in ("summ", "hdr", "boards", "sims", "rootn", "pol", "finals"):
        a = pk[k]
        pad_shape = (max_n,) + ((max_p,) + a.shape[2:] if a.ndim >= 3 else a.shape[1:])
        buf = np.zeros(pad_shape, dtype=a.dtype)
        buf[tuple(slice(0, s) for s in a.shape)] = a
        view = buf.view(np.int64) if buf.dtype == np.uint64 else buf
        t = torch.from_numpy(view).to(dev)
        lst = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
        dist.gather(t, lst, dst=0)
        if rank == 0:
            gathered[k] = [x.cpu().numpy().view(a.dtype) for x in lst]
    if rank != 0:
        return []
    out = []
    for r in range(world):
        n_r, p_r = int(shapes[r][0]), int(shapes[r][1])
        part = {k: gathered[k][r][:n_r] for k in gathered}
        out += unpack_records(part)
    return out


def load_model(config):
    """MultiProcessReversiModelAPIServer.load_model (agent/api.py:102-115): the newest next-generation model or the best
    model, in the order play.use_newest_next_generation_model asks for, unless opts.new; a fresh net is built - and saved
    as the best model, so that the reference's opt / eval workers find one - when nothing could be loaded."""
    from ..agent.model import ReversiModel
    from ..lib.model_helpler import (load_best_model_weight, reload_newest_next_generation_model_if_changed,
                                     save_as_best_model)
    model = ReversiModel(config)
    loaded = False
    if not getattr(getattr(config, "opts", None), "new", False):
        if config.play.use_newest_next_generation_model:
            loaded = reload_newest_next_generation_model_if_changed(model) or load_best_model_weight(model)
        else:
            loaded = load_best_model_weight(model) or reload_newest_next_generation_model_if_changed(model)
    if not loaded:
        model.build()
        save_as_best_model(model)
    return model


def try_reload_model(config, model):
    """try_reload_model (agent/api.py:117-125), polled between batches instead of every 60 s: True when the weight
    file's digest changed and the new weights were loaded."""
    from ..lib.model_helpler import reload_best_model_weight_if_changed, reload_newest_next_generation_model_if_changed
    try:
        if config.play.use_newest_next_generation_model:
            return bool(reload_newest_next_generation_model_if_changed(model))
        return bool(reload_best_model_weight_if_changed(model))
    except Exception as e:   # the reference logs and keeps serving the old weights
        logger.error(e)
        return False


def whole_files_block(config, block_games):
    """block_games rounded up to a whole number of play files and GGF files (nb_game_in_file, nb_game_in_ggf_file), so that under
    several ranks every rank writes the files of its own id range (BatchedSelfPlayWorker._per_rank_emission)."""
    import math
    pd = config.play_data
    unit = max(1, int(pd.nb_game_in_file))
    if pd.enable_ggf_data:
        unit = math.lcm(unit, max(1, int(pd.nb_game_in_ggf_file)))
    return -(-int(block_games) // unit) * unit


def default_block_games(net_blob, games_in_flight):
    """Game ids per rank between two gathers when the caller names none.  A block ends with a ramp-down - its last games finish while
    the slots they leave stay empty - worth about half a game per slot, plus fixed host work (outbox, gather, bookkeeping): at g games
    per slot a block runs at ~ g / (g + 0.5) of the steady state.  Wide nets (a 256x10 game takes ~13 minutes of an 8192-slot batch):
    4 games per slot, so that files and the optimizer's new weights (polled between blocks) turn around within the hour.  Narrow nets
    (a 16-filter block of 4 x 4096 ids lasts 2.4 s): 16 games per slot - 9 s per block, ramp-down and per-block work a quarter of what
    they were (bench.py worker_end_to_end_config1: end to end 91-93 % of the engine-level rate at 4 games per slot)."""
    import struct
    filters = struct.unpack_from("<5i", net_blob, 0)[2] if net_blob and len(net_blob) >= 20 else 256
    return (4 if filters >= 128 else 16) * int(games_in_flight)


def start(config, net_blob=None, games_in_flight=None, total_games=None, seed=0, block_games=None, fused_tree_net="auto", emission="auto"):
    """Reference entry point (worker/self_play.py:28).  Under torchrun uses one rank per GPU.  fused_tree_net: see
    BatchedSelfPlayWorker (16-filter nets: tree and net in one kernel; opt-in).  block_games (ids per rank between two gathers):
    default_block_games - 4 games per slot (16 for narrow nets) -, under several ranks rounded up to whole files (whole_files_block) so
    that each rank writes its own."""
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise RuntimeError("the self-play engine is device-only: no GPU visible (there is no CPU fallback)")
    torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    reload_model = None
    if net_blob is None:
        if rank == 0:
            config.resource.create_directories()
            model = load_model(config)       # may write the fresh best model: one rank only
        if world > 1:
            dist.barrier()
        if rank != 0:
            model = load_model(config)
        net_blob = model.model.to_blob()

        def reload_model():
            return model.model.to_blob() if try_reload_model(config, model) else None
    B = games_in_flight or 4096
    if not block_games:
        block_games = default_block_games(net_blob, B)   # continuous batching: several games per slot between two gathers
        if world > 1 and emission != "rank0":
            block_games = whole_files_block(config, block_games)
    w = BatchedSelfPlayWorker(config, net_blob, B, seed=seed, device=f"cuda:{local}", rank=rank, world=world,
                              block_games=block_games, fused_tree_net=fused_tree_net, emission=emission)
    w.run(total_games, reload_model=reload_model)
    return w
