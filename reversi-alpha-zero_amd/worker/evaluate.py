"""The `eval` worker as ONE batched match on the device — replaces reversi_zero/worker/evaluate.py:17-124.

The reference pits the best model against a challenger one game at a time (evaluate.py:44-96): two fresh
`ReversiPlayer`s per game (separate trees, `config.eval.play_config`), colours drawn at random, and stops early once the
result is decided.  Here all `eval.game_num` games of a challenge are in flight at once:

  * two engines, one per model (an engine evaluates leaves with ONE net): slot g of the best model's engine holds the best
    model's tree of game g, slot g of the challenger's engine the challenger's tree;
  * a round = every unfinished game makes one ply: the mover's engine is armed on the position
    (`raz_engine_set_position`, the engine-side form of `ReversiPlayer.action`), both engines are stepped until every armed
    slot has decided - the searches of all games whose mover is the same model share that model's leaf batches - and the
    actions are applied to the games' `ReversiEnv`s;
  * the reference's early-stopping rule (evaluate.py:55-60) is then applied to the outcomes in game order, which yields
    the decision the sequential loop would have reached (the rule only looks at a prefix of the games).

Each (model, game) pair owns one raz-rng-v1 stream, so a match is reproducible and game g is the game two `ReversiPlayer`
objects on those streams would play (`play_game_sequential`, kept as the cross-check the tests use).
Model-file handling (`load_best_model`, `load_next_generation_model`, `remove_model`) follows evaluate.py:98-124.
"""
import copy
import os
from logging import getLogger
from random import Random
from time import sleep
from types import SimpleNamespace

from ..agent.model import ReversiModel
from ..env.reversi_env import Player, ReversiEnv, Winner
from ..lib.data_helper import get_next_generation_model_dirs
from ..lib.model_helpler import load_best_model_weight, save_as_best_model

logger = getLogger(__name__)


def start(config, max_models=None, seed=0):
    """evaluate.py:17-19.  max_models: stop after that many challengers (None = keep polling, as the reference)."""
    return EvaluateWorker(config, seed=seed).start(max_models=max_models)


def decide(outcomes, game_num, replace_rate):
    """evaluate.py:46-64 on a list of per-game outcomes (1 challenger won, 0 lost, None draw) in game order:
    returns (challenger_is_better, games_the_sequential_loop_would_have_played, winning_rate)."""
    wins = losses = played = 0
    for o in outcomes[:game_num]:
        played += 1
        wins += o == 1
        losses += o == 0
        if losses >= game_num * (1 - replace_rate) or wins >= game_num * replace_rate:
            break
    rate = wins / (wins + losses) if wins + losses else 0
    return rate >= replace_rate, played, rate


class _Side:
    """One model's half of the match: an engine whose slot g is that model's player in game g."""

    def __init__(self, config, model, n_games, seed, first_game_id, device):
        from ..engine import DeviceNet, SelfPlayEngine
        from ..agent.player import effective_play_config
        pc = effective_play_config(config, config.eval.play_config)   # five settings come from config.play whatever play_config says
        # (every ReversiPlayer of evaluate.py:69-70 has a tree of its own: here one engine slot per player, used by that
        #  player only, with the mirror-key updates of player.py:279-280 as the play config's tree mode implies)
        self.sims = int(pc.simulation_num_per_move)
        net = getattr(model, "model", model)
        # "auto": wide nets on the split-f16 trunk (raznet-forward-v2), narrow nets on the exact-f32 kernels (engine.DeviceNet)
        self.engine = SelfPlayEngine(SimpleNamespace(play=pc, play_data=config.play_data),
                                     DeviceNet(net.to_blob(), device, kernel=getattr(config, "net_kernel", "auto")),
                                     n_games, seed=seed, sims_hint=self.sims, max_plies=64, parts=1, single_stream=True)
        self.engine.start(first_game_id, self.sims, n_active=0)   # every slot idle until it is asked for a move
        self.armed = []

    def ask(self, slot, own, enemy):
        # the player's own discs are "black" inside its tree, whatever its colour in the game (agent/player.py:95)
        self.engine.set_position(slot, own, enemy, 1, self.sims, enable_resign=True, one_move=True)
        self.armed.append(slot)

    def busy(self):
        eng = self.engine
        st = eng.stats()   # raises on engine error flags
        cap = int(eng.cfg.nodes_per_game)
        if eng.pool_nearly_full(st, 64):
            eng.gc(threshold=min(cap // 4, st["max_pool_used"] // 2))
        return st["idle_or_done"] < eng.n_games

    def answers(self):
        """{slot: (action or None (resigned), the root's visit counts after the search)} of the slots armed in this round."""
        pk = self.engine.pack_records(0, self.engine.n_games, plies=1)
        from ..engine import PLY_HEADER
        hdr = pk["headers"].cpu().numpy().view(PLY_HEADER).reshape(-1)
        root_n = pk["root_n"].cpu().numpy().reshape(self.engine.n_games, -1)[:, :64]
        out = {g: ((int(hdr[g]["action"]) if int(hdr[g]["action"]) >= 0 else None), root_n[g].copy()) for g in self.armed}
        self.armed = []
        return out


class EvaluateWorker:
    def __init__(self, config, seed=0, device="cuda:0"):
        self.config = config
        self.best_model = None
        self.seed = seed
        self.device = device
        self.random = Random(seed)
        self.games_played = 0

    # -- the challenge loop (evaluate.py:31-42) ---------------------------------------------------------------------
    def start(self, max_models=None):
        self.best_model = self.load_best_model()
        done = 0
        while max_models is None or done < max_models:
            ng_model, model_dir = self.load_next_generation_model()
            logger.debug(f"start evaluate model {model_dir}")
            if self.evaluate_model(ng_model):
                logger.debug(f"New Model become best model: {model_dir}")
                save_as_best_model(ng_model)
                self.best_model = ng_model
            self.remove_model(model_dir)
            done += 1
        return done

    def evaluate_model(self, ng_model):
        ec = self.config.eval
        results = self.play_games(self.best_model, ng_model, ec.game_num)
        better, played, rate = decide([r[0] for r in results], ec.game_num, ec.replace_rate)
        self.last_results = results[:played]
        logger.debug(f"{played} of {ec.game_num} games decide: winning rate {rate * 100:.1f}% -> "
                     f"{'change best model' if better else 'keep best model'}")
        return better

    # -- the batched match ------------------------------------------------------------------------------------------
    def play_games(self, best_model, ng_model, n):
        """n games, all in flight at once.  Returns per game (ng_win: 1 / 0 / None for a draw, best_is_black, (blacks, whites))
        - what the reference's play_game returns (evaluate.py:66-96) - in game order."""
        first = self.games_played
        self.games_played += n
        best_is_black = [self.random.random() < 0.5 for _ in range(n)]
        sides = {True: _Side(self.config, best_model, n, 2 * self.seed, first, self.device),        # the best model
                 False: _Side(self.config, ng_model, n, 2 * self.seed + 1, first, self.device)}     # the challenger
        envs = [ReversiEnv().reset() for _ in range(n)]
        live = list(range(n))
        # what the match looked like, ply by ply (tests compare it with games of the reference's evaluate.py:66-96)
        self.last_games = [{"game_id": first + g, "best_is_black": best_is_black[g], "plies": []} for g in range(n)]
        while live:
            for g in live:
                env = envs[g]
                own, enemy = env.get_own_and_enemy()
                mover_is_best = (env.next_player == Player.black) == best_is_black[g]
                sides[mover_is_best].ask(g, own, enemy)
            stepping = [s for s in sides.values() if s.armed]
            while stepping:
                for s in stepping:
                    s.engine.step(64)
                # every side's busy() runs each round (it also does the side's pool and error-flag checks); a side whose
                # slots have all decided is no longer stepped
                stepping = [s for s, b in [(s, s.busy()) for s in stepping] if b]
            for is_best, s in sides.items():
                for g, (action, root_n) in s.answers().items():
                    self.last_games[g]["plies"].append({"who": "best" if is_best else "ng", "action": -1 if action is None else action, "root_n": root_n})
                    envs[g].step(action)
            live = [g for g in live if not envs[g].done]
        out = []
        for g, env in enumerate(envs):
            ng_win = None
            if env.winner in (Winner.black, Winner.white):
                ng_win = int((env.winner == Winner.black) != best_is_black[g])
            out.append((ng_win, best_is_black[g], env.observation.number_of_black_and_white))
        return out

    def play_game(self, best_model, ng_model):
        """evaluate.py:66-96: one game (a match of one)."""
        return self.play_games(best_model, ng_model, 1)[0]

    def play_game_sequential(self, best_model, ng_model, game_index, best_is_black):
        """Game `game_index` of a match the way evaluate.py:66-96 plays it - two ReversiPlayer objects and one ReversiEnv,
        one move at a time - on the same random streams as the batched match: the cross-check of play_games."""
        from ..agent.player import ReversiPlayer
        pc = self.config.eval.play_config
        players = {True: ReversiPlayer(self.config, best_model, play_config=pc,
                                       mtcs_info=ReversiPlayer.create_mtcs_info(2 * self.seed, game_index, self.device)),
                   False: ReversiPlayer(self.config, ng_model, play_config=pc,
                                        mtcs_info=ReversiPlayer.create_mtcs_info(2 * self.seed + 1, game_index, self.device))}
        env = ReversiEnv().reset()
        while not env.done:
            own, enemy = env.get_own_and_enemy()
            env.step(players[(env.next_player == Player.black) == best_is_black].action(own, enemy))
        ng_win = None
        if env.winner in (Winner.black, Winner.white):
            ng_win = int((env.winner == Winner.black) != best_is_black)
        return ng_win, best_is_black, env.observation.number_of_black_and_white

    # -- model files (evaluate.py:98-124) -----------------------------------------------------------------------------
    def load_best_model(self):
        model = ReversiModel(self.config)
        if not load_best_model_weight(model):
            raise RuntimeError("Best model can not loaded!")
        return model

    def load_next_generation_model(self, poll_seconds=60):
        rc = self.config.resource
        dirs = get_next_generation_model_dirs(rc)
        while not dirs:
            logger.info("There is no next generation model to evaluate")
            sleep(poll_seconds)
            dirs = get_next_generation_model_dirs(rc)
        model_dir = dirs[-1] if self.config.eval.evaluate_latest_first else dirs[0]
        model = ReversiModel(self.config)
        model.load(os.path.join(model_dir, rc.next_generation_model_config_filename),
                   os.path.join(model_dir, rc.next_generation_model_weight_filename))
        return model, model_dir

    def remove_model(self, model_dir):
        rc = self.config.resource
        for name in (rc.next_generation_model_config_filename, rc.next_generation_model_weight_filename):
            os.remove(os.path.join(model_dir, name))
        os.rmdir(model_dir)
