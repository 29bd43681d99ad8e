"""ref_selfplay.py — drive the UNMODIFIED reference self-play (worker/self_play.py
SelfPlayWorker.start_game -> agent/player.py ReversiPlayer -> env/reversi_env.py -> lib/bitboard.py)
with (a) the counter-based stream raz-rng-v1 injected at the reference's random call sites and
(b) the raznet-forward-v1 net plugged in through the ReversiPlayer(api=...) seam (player.py:41).

TEST INFRASTRUCTURE, build-container only (needs /root/reference).  Used by
tests/golden/make_golden_mcts.py and by `needs_reference` tests.  Injection points (module-level
names, no reference file is edited — SURVEY.md §7 hard part 2):
    reversi_zero.agent.player.random        (player.py:8, used :300-301)   -> EXPAND pair
    numpy.random.choice                     (player.py:112)                -> CHOICE
    numpy.random.dirichlet                  (lib/bitboard.py:164)          -> DIRICHLET
    reversi_zero.worker.self_play.random    (self_play.py:6, used :144)    -> GAME d0
    numpy.random.random                     (self_play.py:182)             -> GAME d1
"""
import contextlib
import ctypes
import json
import os
import tempfile

import numpy as np

import oracle as O
import ref_harness as rh


class GameStream:
    """Per-game event counters of raz-rng-v1 (mirrors the fields the engine keeps per game)."""

    def __init__(self, seed, game_id):
        self.seed, self.game_id = seed, game_id
        self.ev_expand = self.ev_choice = self.ev_dirichlet = 0
        self._pending = None

    def expand_uniform(self):
        if self._pending is None:
            d0, d1 = O.rng_pair(self.seed, self.game_id, 0, self.ev_expand)
            self.ev_expand += 1
            self._pending = d1
            return d0
        d1, self._pending = self._pending, None
        return d1

    def choice(self, a, p=None):
        d0, _ = O.rng_pair(self.seed, self.game_id, 1, self.ev_choice)
        self.ev_choice += 1
        cdf = np.asarray(p, dtype=np.float64).cumsum()   # numpy's own recipe (mtrand choice):
        cdf /= cdf[-1]                                    # cdf /= cdf[-1]; searchsorted(side='right')
        return list(a)[int(cdf.searchsorted(d0, side="right"))]

    def dirichlet(self, alpha):
        lib = O.load_ext()
        ev = self.ev_dirichlet
        self.ev_dirichlet += 1
        alpha = list(alpha)
        assert all(a == alpha[0] for a in alpha)
        g = (ctypes.c_double * len(alpha))()
        lib.orc_dirichlet_gammas(float(alpha[0]), len(alpha), self.seed, self.game_id, ev, g)
        acc = 0.0
        for x in g:
            acc += x
        return np.array([x / acc for x in g])

    def game_pair(self):
        return O.rng_pair(self.seed, self.game_id, 3, 0)


class OracleNetAPI:
    """ReversiModelAPI stand-in (agent/api.py:20-45 contract) backed by oracle/orc_net.c."""

    def __init__(self, blob):
        self.blob = blob
        self.calls = 0
        self.positions = 0

    def predict(self, x):
        x = np.asarray(x)
        assert x.ndim in (3, 4)
        single = x.ndim == 3
        pol, val = O.net_forward_planes(self.blob, x.reshape(-1, 2, 8, 8))
        self.calls += 1
        self.positions += pol.shape[0]
        return (pol[0], val[0]) if single else (pol, val)


@contextlib.contextmanager
def injected(stream):
    rh.install()
    import reversi_zero.agent.player as rp
    import reversi_zero.worker.self_play as rsp
    saved = (rp.random, np.random.choice, np.random.dirichlet, rsp.random, np.random.random, rp.ReversiPlayer.__init__)
    orig_init = rp.ReversiPlayer.__init__

    def init_with_sem(self, *a, **k):
        orig_init(self, *a, **k)
        self.sem = rh.CompatSemaphore(self.play_config.parallel_search_num)  # player.py:205 on py>=3.9

    rp.random = stream.expand_uniform
    np.random.choice = stream.choice
    np.random.dirichlet = stream.dirichlet
    rsp.random = lambda: stream.game_pair()[0]
    np.random.random = lambda *a, **k: stream.game_pair()[1]
    rp.ReversiPlayer.__init__ = init_with_sem
    try:
        yield
    finally:
        (rp.random, np.random.choice, np.random.dirichlet, rsp.random, np.random.random,
         rp.ReversiPlayer.__init__) = saved


def run_reference_game(config, blob, seed, game_id, sims_per_move, data_dir=None, virtual_time=False, carry=None, api=None, start=None):
    """One game through the reference's SelfPlayWorker.start_game.  Returns a dict with per-ply
    captures (root N/W, action, n, q, emitted rows) and the play_*.json content the reference wrote.
    carry (dict, optional): holds the worker's MCTSInfo from call to call, the way SelfPlayWorker.start
    keeps `mtcs_info` for reset_mtcs_info_per_game games (worker/self_play.py:109-111,132-134).
    start (optional): (black, white, next_player) - the worker's env is put on that position by ReversiEnv.update
    (env/reversi_env.py:33-40) where start_game resets it (worker/self_play.py:143), i.e. the game is taken up there."""
    rh.install()
    import reversi_zero.agent.player as rp
    from reversi_zero.env.reversi_env import ReversiEnv, Player
    from reversi_zero.worker.self_play import SelfPlayWorker
    own_tmp = data_dir is None
    data_dir = data_dir or tempfile.mkdtemp(prefix="raz_ref_")
    rc = config.resource
    rc.data_dir = data_dir
    rc.play_data_dir = os.path.join(data_dir, "play_data")
    rc.self_play_ggf_data_dir = os.path.join(data_dir, "self_play-ggf")
    rc.force_simulation_num_file = os.path.join(data_dir, ".force-sim")
    rc.self_play_game_idx_file = os.path.join(data_dir, ".self-play-game-idx")
    os.makedirs(rc.play_data_dir, exist_ok=True)
    os.makedirs(rc.self_play_ggf_data_dir, exist_ok=True)
    config.play.schedule_of_simulation_num_per_move = [(0, sims_per_move)]
    config.play_data.nb_game_in_file = 1
    stream = GameStream(seed, game_id)
    api = api if api is not None else OracleNetAPI(blob)
    plies = []
    orig_awe = rp.ReversiPlayer.action_with_evaluation

    def capture(self, own, enemy, callback_in_mtcs=None):
        n_rows = len(self.moves)
        res = orig_awe(self, own, enemy, callback_in_mtcs=callback_in_mtcs)
        key = rp.CounterKey(own, enemy, Player.black.value)
        new = self.moves[n_rows:]
        plies.append({"own": own, "enemy": enemy, "action": -1 if res.action is None else int(res.action),
                      "n": float(res.n), "q": float(res.q),
                      "root_n": [float(v) for v in self.var_n[key]], "root_w": [float(v) for v in self.var_w[key]],
                      "has_row": len(new) == 8,
                      "solved": (len(new) == 0 and res.action is not None and float(res.n) == 999.0
                                 and not isinstance(res.q, float)),
                      "saved_policy": [float(v) for v in new[0][1]] if new else None})
        return res

    old_loop = None
    if virtual_time:   # parallel_search_num > 1: the deterministic stage of raz-sched-v1 (ref_harness.VirtualTimeLoop)
        import asyncio
        old_loop = asyncio.get_event_loop_policy().get_event_loop()
        asyncio.set_event_loop(rh.VirtualTimeLoop())
    with injected(stream):
        rp.ReversiPlayer.action_with_evaluation = capture
        try:
            the_env = ReversiEnv()
            if start is not None:
                the_env.reset = lambda: the_env.update(int(start[0]), int(start[1]), Player(int(start[2])))
            worker = SelfPlayWorker(config, env=the_env, api=api, shared_var=None, worker_index=0)
            mtcs_info = carry.get("mtcs_info") if carry is not None else None
            if mtcs_info is None and config.play.share_mtcs_info_in_self_play:   # self_play.py:109-111
                mtcs_info = rp.ReversiPlayer.create_mtcs_info()
            if carry is not None:
                carry["mtcs_info"] = mtcs_info
            env = worker.start_game(1, 0, mtcs_info)
        finally:
            rp.ReversiPlayer.action_with_evaluation = orig_awe
            if old_loop is not None:
                import asyncio
                asyncio.get_event_loop().close()
                asyncio.set_event_loop(old_loop)
    files = sorted(os.listdir(rc.play_data_dir))
    play_rows = None
    if files:
        with open(os.path.join(rc.play_data_dir, files[-1])) as f:
            play_rows = json.load(f)
    players = []
    e2 = ReversiEnv().reset() if start is None else ReversiEnv().update(int(start[0]), int(start[1]), Player(int(start[2])))
    for p in plies:  # recover who moved at each ply by replaying
        players.append(e2.next_player.value)
        e2.step(None if p["action"] < 0 else p["action"])
    for p, who in zip(plies, players):
        p["player"] = who
    out = {"seed": seed, "game_id": game_id, "sims_per_move": sims_per_move, "plies": plies,
           "winner": env.winner.value, "turn": env.turn, "black": env.board.black, "white": env.board.white,
           "resigned_black": bool(worker.black.resigned), "resigned_white": bool(worker.white.resigned),
           "nn_positions": api.positions, "play_rows": play_rows,
           "ggf_moves": list(worker.move_history.moves),
           "mirror_or_total_keys": len(worker.black.var_n)}
    if own_tmp:
        import shutil
        shutil.rmtree(data_dir, ignore_errors=True)
    return out
