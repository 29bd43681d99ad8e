"""Driver of the WAVE-EMULATOR build of the tree kernels (tests/native/wave_emu -> tests/native/libraz_emu.so): the engine's
C ABI (include/raz.h) on host memory, so that csrc/raz_engine.hip can be stepped and checked against the CPU oracle in a
container without a GPU.  TEST INFRASTRUCTURE ONLY - the product never loads this library, and the GPU parity tests
(tests/test_engine_gpu.py ...) remain the tests of record; this is the debugging loop the build container otherwise lacks."""
import ctypes
import os
import subprocess

import numpy as np

from conftest import ROOT, _locked

EMU_DIR = os.path.join(ROOT, "tests", "native", "wave_emu")
EMU_LIB = os.path.join(ROOT, "tests", "native", "libraz_emu.so")
EMU_FULL_LIB = os.path.join(ROOT, "tests", "native", "libraz_emu_full.so")
_libs = {}


def load(full=False):
    """full=False: the tree kernels, leaves evaluated by the oracle's C net (libraz_emu.so, seconds to build).  full=True: the
    whole product incl. the real net kernels on emulated matrix cores (libraz_emu_full.so, a minute to build): what the fused
    tree + net kernel needs."""
    if full not in _libs:
        # RAZ_EMU_VARIANT=<name> RAZ_EMU_EXTRA=<flags>: a build of the tree kernels with other compile-time options, in a library of its own
        # (a child process of a test that wants the kernels' optional forms checked: tests/test_engine_emu.py)
        variant, extra = os.environ.get("RAZ_EMU_VARIANT"), os.environ.get("RAZ_EMU_EXTRA", "")
        path = EMU_FULL_LIB if full else (os.path.join(ROOT, "tests", "native", f"libraz_emu_{variant}.so") if variant else EMU_LIB)
        args = ["../libraz_emu_full.so"] if full else ([f"OUT=../libraz_emu_{variant}.so", f"EXTRA={extra}"] if variant else [])
        with _locked("emu" + (f"_{variant}" if variant and not full else "")):
            r = subprocess.run(["make", "-C", EMU_DIR] + args, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("wave-emulator build failed:\n" + r.stdout[-3000:] + r.stderr[-3000:])
        from reversi_alpha_zero_amd import _native as N
        lib = ctypes.CDLL(path)
        lib.raz_last_error.restype = ctypes.c_char_p
        for name, (res, args) in N.SIGNATURES.items():
            fn = getattr(lib, name, None)
            if fn is not None:
                fn.restype, fn.argtypes = res, args
        _libs[full] = lib
    return _libs[full]


def _check(lib, rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed (code {rc}): {(lib.raz_last_error() or b'').decode()}")


class EmuEngine:
    """The subset of reversi_alpha_zero_amd.engine.SelfPlayEngine the parity tests use, on the emulated kernels."""

    def __init__(self, config, blob, n_games, seed=0, nodes_per_game=None, sims_hint=None, max_plies=72, record_root_w=True,
                 inner_max=0, force_slot_kernel=False, pool_bytes_per_game=0, fused=False, full_lib=False, **cfg_overrides):
        """fused: the tree + net kernel (raz_engine_config.reserved bit 4); it, and full_lib, run on the whole-product library."""
        from reversi_alpha_zero_amd import _native as N
        from reversi_alpha_zero_amd.engine import engine_config_from
        self.lib = lib = load(full=bool(fused or full_lib))
        self.n_games, self.max_plies, self.record_root_w = n_games, max_plies, record_root_w
        import struct
        _, _, F, R, V = struct.unpack_from("<5i", blob, 0)
        self._weights = np.zeros(lib.raz_net_weight_bytes(F, R, V), dtype=np.uint8)
        self.net = N.RazNet()
        _check(lib, lib.raz_net_load(ctypes.byref(self.net), blob, len(blob), self._weights.ctypes.data, self._weights.size, None), "raz_net_load")
        if nodes_per_game is None:
            s = sims_hint or config.play.simulation_num_per_move
            share = bool(config.play.share_mtcs_info_in_self_play)
            nodes_per_game = (s * max(1, config.play.thinking_loop) * 62 + 128) * (2 if share else 1)
        from reversi_alpha_zero_amd.engine import NODE_MAX_BYTES, NODE_POOL_BYTES_PER_NODE
        self.cfg = engine_config_from(config, n_games, seed, nodes_per_game, max_plies, None, record_root_w, False, True, 1, inner_max,
                                      force_slot_kernel=force_slot_kernel, pool_bytes_per_game=pool_bytes_per_game, fused=fused)
        self.pool_bytes = int(pool_bytes_per_game) or (int(nodes_per_game) * NODE_POOL_BYTES_PER_NODE + 64 * NODE_MAX_BYTES)
        self.cfg.reserved |= (int(cfg_overrides.pop("solver_pool_every", 0)) & 0xf) << 24   # tree launches per round of the solver pool
        for k, v in cfg_overrides.items():
            setattr(self.cfg, k, v)
        self.slots = int(self.cfg.parallel_search_num) or 1
        nbytes = lib.raz_engine_workspace_bytes(ctypes.byref(self.cfg))
        if nbytes == 0:
            raise ValueError("invalid engine config: " + (lib.raz_last_error() or b"").decode())
        self.workspace_bytes = nbytes
        self._ws = np.zeros(nbytes + 256, dtype=np.uint8)
        base = (self._ws.ctypes.data + 255) // 256 * 256
        self._h = ctypes.c_void_p()
        need = lib.raz_net_scratch_bytes(F, V, n_games * self.slots)
        self._scratch = np.zeros(max(need, 8), dtype=np.uint8)
        _check(lib, lib.raz_engine_create(ctypes.byref(self.cfg), ctypes.byref(self.net), base, nbytes, self._scratch.ctypes.data if need else None, need,
                                          ctypes.byref(self._h)), "raz_engine_create")
        self.nodes_per_step = 2 * (inner_max or 2) if (self.slots == 1 and not force_slot_kernel) else 3 * self.slots + (inner_max or 2)

    def __del__(self):
        if getattr(self, "_h", None):
            self.lib.raz_engine_destroy(self._h)
            self._h = None

    def start(self, first_game_id, sims_per_move, n_active=None):
        sims = np.full(self.n_games, sims_per_move, dtype=np.uint32) if np.isscalar(sims_per_move) else np.ascontiguousarray(sims_per_move, dtype=np.uint32)
        self.n_active = self.n_games if n_active is None else n_active
        _check(self.lib, self.lib.raz_engine_start(self._h, first_game_id, sims.ctypes.data, self.n_active, None), "raz_engine_start")

    def next_game(self, first_game_id, sims_per_move, n_active=None):
        sims = np.full(self.n_games, sims_per_move, dtype=np.uint32)
        self.n_active = self.n_games if n_active is None else n_active
        _check(self.lib, self.lib.raz_engine_next_game(self._h, first_game_id, sims.ctypes.data, self.n_active, None), "raz_engine_next_game")

    def step(self, n=1):
        _check(self.lib, self.lib.raz_engine_step(self._h, n, None), "raz_engine_step")

    def gc(self, threshold=0):
        _check(self.lib, self.lib.raz_engine_gc(self._h, threshold, None), "raz_engine_gc")

    def stats(self):
        from reversi_alpha_zero_amd import _native as N
        st = N.RazEngineStats()
        _check(self.lib, self.lib.raz_engine_stats_sync(self._h, ctypes.byref(st), None), "raz_engine_stats_sync")
        if st.error_flags:
            raise RuntimeError(f"engine error flags {st.error_flags:#x}")
        return {"finished_games": st.finished_games, "total_sims": st.total_sims, "nn_leaves": st.nn_leaves, "selections": st.selections,
                "max_pool_used": st.max_pool_used, "idle_or_done": st.idle_or_done, "max_pool_bytes": st.max_pool_bytes}

    def pool_nearly_full(self, st, steps):
        from reversi_alpha_zero_amd.engine import SelfPlayEngine
        return SelfPlayEngine.pool_nearly_full(self, st, steps)

    def run(self, chunk=16, max_steps=200000, allow_gc=True):
        steps, cap, self.gc_runs = 0, int(self.cfg.nodes_per_game), 0
        while True:
            self.step(chunk)
            steps += chunk
            st = self.stats()
            if allow_gc and self.pool_nearly_full(st, chunk):
                self.gc(min(cap // 4, st["max_pool_used"] // 2))
                self.gc_runs += 1
            if st["finished_games"] >= self.n_active:
                st["steps"] = steps
                return st
            if steps >= max_steps:
                raise RuntimeError("engine did not finish")

    def set_position(self, slot, black, white, player, sims, enable_resign=True, one_move=True):
        _check(self.lib, self.lib.raz_engine_set_position(self._h, slot, black, white, player, sims, int(enable_resign), int(one_move), None), "raz_engine_set_position")

    def set_positions(self, first_slot, black, white, player, sims, enable_resign=True, one_move=True):
        b, w, p = (np.ascontiguousarray(black, dtype=np.uint64), np.ascontiguousarray(white, dtype=np.uint64), np.ascontiguousarray(player, dtype=np.uint8))
        _check(self.lib, self.lib.raz_engine_set_positions(self._h, first_slot, b.size, b.ctypes.data, w.ctypes.data, p.ctypes.data, sims,
                                                           int(enable_resign), int(one_move), None), "raz_engine_set_positions")

    def stop_thinking(self, slot):
        _check(self.lib, self.lib.raz_engine_stop_thinking(self._h, slot, None), "raz_engine_stop_thinking")

    def adopt_tree(self, slot, player_index):
        _check(self.lib, self.lib.raz_engine_adopt_tree(self._h, slot, int(player_index), None), "raz_engine_adopt_tree")

    def read_node(self, slot, black, white, next_player=1, owner=0):
        w, n, p = np.zeros(64), np.zeros(64, dtype=np.uint32), np.zeros(64, dtype=np.float32)
        found = ctypes.c_int(0)
        _check(self.lib, self.lib.raz_engine_read_node(self._h, slot, black, white, next_player, owner, w.ctypes.data, n.ctypes.data, p.ctypes.data,
                                                       ctypes.byref(found), None), "raz_engine_read_node")
        return bool(found.value), w, n, p

    def read_raw(self):
        from reversi_alpha_zero_amd.engine import PLY_HEADER
        B, MP = self.n_games, self.max_plies
        hdr = np.zeros((B, MP), dtype=PLY_HEADER)
        root_n = np.zeros((B, MP, 64), dtype=np.uint32)
        root_w = np.zeros((B, MP, 64), dtype=np.float64) if self.record_root_w else None
        n_plies, status, resigned = np.zeros(B, dtype=np.uint32), np.zeros(B, dtype=np.uint8), np.zeros((B, 2), dtype=np.uint8)
        game_id, enable_resign = np.zeros(B, dtype=np.uint32), np.zeros(B, dtype=np.uint8)
        fb, fw = np.zeros(B, dtype=np.uint64), np.zeros(B, dtype=np.uint64)
        _check(self.lib, self.lib.raz_engine_read_records(self._h, hdr.ctypes.data, root_n.ctypes.data, root_w.ctypes.data if root_w is not None else None,
                                                          n_plies.ctypes.data, status.ctypes.data, resigned.ctypes.data, game_id.ctypes.data,
                                                          enable_resign.ctypes.data, fb.ctypes.data, fw.ctypes.data, None), "raz_engine_read_records")
        return dict(headers=hdr, root_n=root_n, root_w=root_w, n_plies=n_plies, status=status, resigned=resigned, game_id=game_id,
                    enable_resign=enable_resign, final_black=fb, final_white=fw)

    # ---- continuous batching (raz_engine_harvest) on host arrays
    def new_outbox(self, first_game_id, n_games):
        mp = self.max_plies
        return {"first": first_game_id, "n": n_games, "headers": np.zeros((n_games, mp, 48), dtype=np.uint8),
                "root_n": np.zeros((n_games, mp, 64), dtype=np.int32), "summary": np.zeros((n_games, 32), dtype=np.uint8),
                "done": np.zeros(n_games, dtype=np.uint8)}

    def harvest(self, outbox, next_game_id, sims_per_move):
        from reversi_alpha_zero_amd import _native as N
        sims = np.ascontiguousarray(sims_per_move, dtype=np.uint32)
        res = N.RazHarvestResult()
        _check(self.lib, self.lib.raz_engine_harvest(self._h, next_game_id, sims.size, sims.ctypes.data if sims.size else None, None,
                                                     outbox["first"], outbox["n"], outbox["headers"].ctypes.data, outbox["root_n"].ctypes.data,
                                                     outbox["summary"].ctypes.data, outbox["done"].ctypes.data, ctypes.byref(res), None),
               "raz_engine_harvest")
        return res.harvested, res.restarted, res.skipped, res.playing

    def play_continuous(self, first_game_id, total_games, sims, chunk=16):
        B = self.n_games
        n0 = min(B, total_games)
        self.start(first_game_id, sims, n_active=n0)
        outbox = self.new_outbox(first_game_id, total_games)
        nxt, done, end, self.gc_runs = first_game_id + n0, 0, first_game_id + total_games, 0
        cap = int(self.cfg.nodes_per_game)
        while done < total_games:
            self.step(chunk)
            st = self.stats()
            if self.pool_nearly_full(st, chunk):
                self.gc(min(cap // 4, st["max_pool_used"] // 2))
                self.gc_runs += 1
            k = min(B, end - nxt)
            h, r, skipped, _ = self.harvest(outbox, nxt, [sims] * k)
            assert skipped == 0
            nxt += r
            done += h
        return outbox

    def attach_leaf_cache(self, log2_entries, max_discs=0):
        need = self.lib.raz_leaf_cache_bytes(log2_entries, self.n_games * self.slots)
        self._cache = np.zeros(need + 256, dtype=np.uint8)
        base = (self._cache.ctypes.data + 255) // 256 * 256
        _check(self.lib, self.lib.raz_engine_set_leaf_cache(self._h, base, need, log2_entries, max_discs, None), "raz_engine_set_leaf_cache")

    def leaf_cache_stats(self):
        out = (ctypes.c_uint64 * 4)()
        _check(self.lib, self.lib.raz_engine_leaf_cache_stats(self._h, out, None), "raz_engine_leaf_cache_stats")
        return {"hits": out[0], "in_batch_duplicates": out[1], "evaluated": out[2], "no_room": out[3]}

    def records(self, save_policy_of_tau_1=True, change_tau_turn=None):
        from reversi_alpha_zero_amd.engine import SelfPlayEngine
        return SelfPlayEngine.records(self, save_policy_of_tau_1, change_tau_turn)
