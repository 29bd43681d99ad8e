"""ReversiPlayer on the device engine — the drop-in for reversi_zero/agent/player.py:28-428.

Same constructor, methods, attributes and return types as the reference class; the search itself
(search_moves / search_my_move / expand_and_evaluate / select_action_q_and_u / the solver calls,
player.py:189-428) runs inside one slot of a `raz_engine` (include/raz.h):

  * `MCTSInfo` (player.py:22,63-66) is a handle to one engine slot — the search tree of one game.
    `var_n[key]`, `var_w[key]`, `var_p[key]` read the slot's statistics of
    `CounterKey(black, white, next_player)` back from HBM (`raz_engine_read_node`).
    Two players built on the same MCTSInfo share the tree exactly as the reference's players share the
    dicts (share_mtcs_info_in_self_play); each keeps its own `expanded` set (a bit per key and player).
  * `action_with_evaluation(own, enemy)` arms one move on the slot (`raz_engine_set_position`, tree
    kept), steps the engine until the move is decided and reads the ply back.  Tree statistics,
    thinking loops, resignation, tau, the solver and the 8 symmetric rows follow the reference; the
    random draws come from the raz-rng-v1 stream of (seed, game_id) given to `create_mtcs_info`
    (DESIGN.md), so a game driven through this class is bit-identical to the same game id inside the
    batched worker and to oracle/orc_mcts.c.

There is no CPU path: without a GPU and libraz.so the constructor raises.
"""
from collections import namedtuple
from logging import getLogger
from types import SimpleNamespace
import copy

import numpy as np

from ..env.reversi_env import Player, ReversiEnv

CounterKey = namedtuple("CounterKey", "black white next_player")
QueueItem = namedtuple("QueueItem", "state future")
HistoryItem = namedtuple("HistoryItem", "action policy values visit enemy_values enemy_visit")
CallbackInMCTS = namedtuple("CallbackInMCTS", "per_sim callback")
ActionWithEvaluation = namedtuple("ActionWithEvaluation", "action n q")

logger = getLogger(__name__)


class _NodeView:
    """dict-like read access `view[CounterKey]` -> np.ndarray(64) on one field of a slot's tree."""

    def __init__(self, info, field):
        self._info, self._field = info, field

    def __getitem__(self, key):
        info = self._info
        if info.engine is None:
            return np.zeros((64,))
        found, w, n, p = info.engine.read_node(info.slot, int(key[0]), int(key[1]), int(key[2]), info.view_owner)
        return {"w": w, "n": n.astype(np.float64), "p": p.astype(np.float64)}[self._field]

    def __contains__(self, key):
        info = self._info
        if info.engine is None:
            return False
        return info.engine.read_node(info.slot, int(key[0]), int(key[1]), int(key[2]), info.view_owner)[0]


class MCTSInfo:
    """The reference's MCTSInfo(var_n, var_w, var_p) (player.py:22) as a device-tree handle.
    Unpacks like the namedtuple: `var_n, var_w, var_p = mtcs_info`."""

    def __init__(self, seed=0, game_id=0, device="cuda:0", nodes=None):
        self.seed, self.game_id, self.device, self.nodes = seed, game_id, device, nodes
        self.engine = None
        self.slot = 0
        self.view_owner = 0
        self.attached = 0
        self.var_n = _NodeView(self, "n")
        self.var_w = _NodeView(self, "w")
        self.var_p = _NodeView(self, "p")

    def __iter__(self):
        return iter((self.var_n, self.var_w, self.var_p))

    def _bind(self, engine_factory):
        if self.engine is None:
            self.engine = engine_factory(self)
            self.engine.start(self.game_id, 1, n_active=0)   # every slot idle; random-stream counters at 0
        idx = self.attached % 2
        if self.attached >= 2:  # a new player on a used tree: expanded = set(var_p.keys()) (player.py:47)
            self.engine.adopt_tree(self.slot, idx)
        self.attached += 1
        return idx


def effective_play_config(config, play_config=None):
    """The settings a reference ReversiPlayer built with `play_config` really plays under.  agent/player.py takes most of them from
    self.play_config (= play_config or config.play, :39) but reads FIVE from the global config.play whatever play_config says:
    allowed_resign_turn (:127), use_solver_turn_in_simulation (:237-238), virtual_loss (:264), policy_decay_turn / _power (:410-411) -
    and the two sleep constants that only order events (:254, :341).  So an `eval` match (config.eval.play_config) resigns, solves
    inside simulations and applies virtual losses by the SELF-PLAY section's values; pinned by tests/golden/eval_games.json."""
    pc = copy.copy(play_config if play_config is not None else config.play)
    g = config.play
    for name in ("allowed_resign_turn", "use_solver_turn_in_simulation", "virtual_loss", "policy_decay_turn", "policy_decay_power"):
        if hasattr(g, name):
            setattr(pc, name, getattr(g, name))
    return pc


class ReversiPlayer:
    def __init__(self, config, model, play_config=None, enable_resign=True, mtcs_info=None, api=None):
        """config: Config; model: agent.model.ReversiModel (or anything with .model.to_blob() / .to_blob());
        play_config: overrides config.play; mtcs_info: MCTSInfo handle (None: a fresh tree);
        api: a ReversiModelAPI whose device weights are reused (the reference's parameter of the same name)."""
        self.config = config
        self.model = model
        self.play_config = play_config or self.config.play
        self.enable_resign = enable_resign
        self.api = api
        if not 1 <= int(getattr(self.play_config, "parallel_search_num", 1) or 1) <= 16:
            raise ValueError("ReversiPlayer: parallel_search_num must be 1..16 (prediction_queue_size, config.py:141)")
        mtcs_info = mtcs_info or self.create_mtcs_info()
        self.mtcs_info = mtcs_info
        self.var_n, self.var_w, self.var_p = mtcs_info
        self.moves = []
        self.callback_in_mtcs = None
        self.thinking_history = {}  # for fun
        self.resigned = False
        self.requested_stop_thinking = False
        self.player_index = mtcs_info._bind(self._make_engine)
        self._share = bool(mtcs_info.engine.cfg.share_mtcs_info)

    # -- engine construction ---------------------------------------------------------------------
    def _make_engine(self, info):
        from ..engine import DeviceNet, SelfPlayEngine
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("ReversiPlayer runs on the device engine: no GPU visible (there is no CPU fallback)")
        if self.api is not None and hasattr(self.api, "_device_net"):
            net = self.api._device_net()
        else:
            m = getattr(self.model, "model", self.model)
            cache = getattr(m, "_raz_device_nets", None)
            if cache is None:
                cache = m._raz_device_nets = {}
            net = cache.get(info.device)
            if net is None:
                net = cache[info.device] = DeviceNet(m.to_blob(), info.device)
        pc = effective_play_config(self.config, self.play_config)
        shim = SimpleNamespace(play=pc, play_data=self.config.play_data)
        sims = int(pc.simulation_num_per_move)
        return SelfPlayEngine(shim, net, 1, seed=info.seed, nodes_per_game=info.nodes, sims_hint=sims,
                              record_root_w=True, single_stream=True, parts=1)

    @staticmethod
    def create_mtcs_info(seed=0, game_id=0, device="cuda:0", nodes=None):
        """player.py:62-66.  seed / game_id select the raz-rng-v1 stream of the tree's random draws."""
        return MCTSInfo(seed, game_id, device, nodes)

    # -- reference API ---------------------------------------------------------------------------
    def var_q(self, key):
        return self.var_w[key] / (self.var_n[key] + 1e-5)

    def action(self, own, enemy, callback_in_mtcs=None):
        """:return action=move pos=0 ~ 63 (0=top left, 7 top right, 63 bottom right), None = resign"""
        return self.action_with_evaluation(own, enemy, callback_in_mtcs=callback_in_mtcs).action

    def action_with_evaluation(self, own, enemy, callback_in_mtcs=None):
        """player.py:82-134.  :rtype: ActionWithEvaluation(action, n=N of the action, q=W/N of the action)"""
        info, eng = self.mtcs_info, self.mtcs_info.engine
        pc = self.play_config
        self.callback_in_mtcs = callback_in_mtcs
        self.requested_stop_thinking = False
        pl = self.player_index
        owner = 0 if self._share else pl
        info.view_owner = owner
        black, white = (own, enemy) if pl == 0 else (enemy, own)
        sims = int(pc.simulation_num_per_move)
        eng.set_position(info.slot, black, white, pl + 1, sims, self.enable_resign, one_move=True)
        key = CounterKey(own, enemy, Player.black.value)
        cap = int(eng.cfg.nodes_per_game)
        chunk = max(1, min(64, sims)) if not callback_in_mtcs else max(1, int(callback_in_mtcs.per_sim))
        stop_sent = False
        while True:
            eng.step(chunk)
            st = eng.stats()   # raises on engine error flags (node pool / table / record overflow)
            if st["idle_or_done"] >= eng.n_games:
                break
            if eng.pool_nearly_full(st, chunk):
                eng.gc(threshold=min(cap // 4, st["max_pool_used"] // 2))
            if callback_in_mtcs and callback_in_mtcs.callback:
                callback_in_mtcs.callback(list(self.var_q(key)), list(self.var_n[key]))
            if self.requested_stop_thinking and not stop_sent:
                eng.stop_thinking(info.slot)
                stop_sent = True
        raw = eng.read_raw()
        assert int(raw["n_plies"][0]) == 1
        h = raw["headers"][0, 0]
        action = int(h["action"])
        if int(h["flags"]) & 1:  # solved: not saved as play data (player.py:100-103,150-161)
            return ActionWithEvaluation(action=action, n=int(h["n"]), q=int(h["q"]))
        n = raw["root_n"][0, 0].astype(np.float64)
        w = raw["root_w"][0, 0]
        q = w / (n + 1e-5)
        turn = int(h["turn"])
        from ..engine import saved_policy
        policy = np.asarray(saved_policy(n, turn, pc.change_tau_turn, False))
        chosen = action if action >= 0 else int(np.argmax(n))
        if action >= 0:
            self.update_thinking_history(own, enemy, chosen, policy, q, n)
        if pc.resign_threshold is not None and np.max(q - (n == 0) * 10) <= pc.resign_threshold:
            self.resigned = True
        if action < 0:
            return ActionWithEvaluation(None, 0, 0)  # means resign
        saved = saved_policy(n, turn, pc.change_tau_turn, self.config.play_data.save_policy_of_tau_1)
        self.add_data_to_move_buffer_with_8_symmetries(own, enemy, saved)
        return ActionWithEvaluation(action=action, n=n[action], q=q[action])

    def update_thinking_history(self, black, white, action, policy, q=None, n=None):
        key = CounterKey(black, white, Player.black.value)
        next_key = self.get_next_key(black, white, action)
        q = self.var_q(key) if q is None else q
        n = self.var_n[key] if n is None else n
        self.thinking_history[(black, white)] = \
            HistoryItem(action, policy, list(q), list(n), list(self.var_q(next_key)), list(self.var_n[next_key]))

    def stop_thinking(self):
        self.requested_stop_thinking = True

    def add_data_to_move_buffer_with_8_symmetries(self, own, enemy, policy):
        from ..worker.self_play import symmetric_rows
        self.moves += symmetric_rows(own, enemy, policy)

    def get_next_key(self, own, enemy, action):
        env = ReversiEnv().update(own, enemy, Player.black)
        env.step(action)
        return self.counter_key(env)

    def ask_thought_about(self, own, enemy) -> HistoryItem:
        return self.thinking_history.get((own, enemy))

    def finish_game(self, z):
        """:param z: win=1, lose=-1, draw=0"""
        for move in self.moves:  # add this game winner result to all past moves.
            move += [z]

    @staticmethod
    def counter_key(env: ReversiEnv):
        return CounterKey(env.board.black, env.board.white, env.next_player.value)

    @staticmethod
    def another_side_counter_key(env: ReversiEnv):
        return CounterKey(env.board.white, env.board.black, 3 - env.next_player.value)

    @staticmethod
    def normalize(p, t=1):
        pp = np.power(p, t)
        return pp / np.sum(pp)
