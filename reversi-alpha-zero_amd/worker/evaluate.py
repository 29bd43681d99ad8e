"""The `eval` worker on the device engine — drop-in for reversi_zero/worker/evaluate.py:17-124.

Best model vs next-generation model: `game_num` games, colours drawn at random, each side a
`ReversiPlayer` (agent/player.py of this package: one engine slot per player, searching with
`config.eval.play_config`); the challenger replaces the best model when its winning rate over the
decided games reaches `replace_rate`, with the reference's early stopping.  Same class, method
names and file handling as the reference, so `manager.py`'s `eval` command only needs the import
changed.  Random draws (colour assignment) come from a seeded `random.Random`, the players'
move sampling from their raz-rng-v1 streams: an evaluation is reproducible.
"""
import os
from logging import getLogger
from random import Random
from time import sleep

from ..agent.model import ReversiModel
from ..agent.player import ReversiPlayer
from ..env.reversi_env import Player, ReversiEnv, Winner
from ..lib.data_helper import get_next_generation_model_dirs
from ..lib.model_helpler import load_best_model_weight, save_as_best_model

logger = getLogger(__name__)


def start(config, max_models=None, seed=0):
    """evaluate.py:17-19.  max_models: stop after that many challengers (None = run forever, as the reference)."""
    return EvaluateWorker(config, seed=seed).start(max_models=max_models)


class EvaluateWorker:
    def __init__(self, config, seed=0, device="cuda:0"):
        self.config = config
        self.best_model = None
        self.seed = seed
        self.device = device
        self.random = Random(seed)
        self.games_played = 0

    def start(self, max_models=None):
        self.best_model = self.load_best_model()
        done = 0
        while max_models is None or done < max_models:
            ng_model, model_dir = self.load_next_generation_model()
            logger.debug(f"start evaluate model {model_dir}")
            ng_is_great = self.evaluate_model(ng_model)
            if ng_is_great:
                logger.debug(f"New Model become best model: {model_dir}")
                save_as_best_model(ng_model)
                self.best_model = ng_model
            self.remove_model(model_dir)
            done += 1
        return done

    def evaluate_model(self, ng_model):
        """evaluate.py:44-64."""
        ec = self.config.eval
        results = []
        winning_rate = 0
        for game_idx in range(ec.game_num):
            # ng_win := if ng_model win -> 1, lose -> 0, draw -> None
            ng_win, black_is_best, black_white = self.play_game(self.best_model, ng_model)
            if ng_win is not None:
                results.append(ng_win)
                winning_rate = sum(results) / len(results)
            logger.debug(f"game {game_idx}: ng_win={ng_win} black_is_best_model={black_is_best} score={black_white} "
                         f"winning rate {winning_rate * 100:.1f}%")
            if results.count(0) >= ec.game_num * (1 - ec.replace_rate):
                logger.debug(f"lose count reach {results.count(0)} so give up challenge")
                break
            if results.count(1) >= ec.game_num * ec.replace_rate:
                logger.debug(f"win count reach {results.count(1)} so change best model")
                break
        winning_rate = sum(results) / len(results) if results else 0
        logger.debug(f"winning rate {winning_rate * 100:.1f}%")
        return winning_rate >= ec.replace_rate

    def play_game(self, best_model, ng_model):
        """evaluate.py:66-96."""
        env = ReversiEnv().reset()
        gid = self.games_played
        self.games_played += 1
        pc = self.config.eval.play_config
        best_player = ReversiPlayer(self.config, best_model, play_config=pc,
                                    mtcs_info=ReversiPlayer.create_mtcs_info(self.seed, 2 * gid, self.device))
        ng_player = ReversiPlayer(self.config, ng_model, play_config=pc,
                                  mtcs_info=ReversiPlayer.create_mtcs_info(self.seed, 2 * gid + 1, self.device))
        best_is_black = self.random.random() < 0.5
        if best_is_black:
            black, white = best_player, ng_player
        else:
            black, white = ng_player, best_player
        observation = env.observation
        while not env.done:
            if env.next_player == Player.black:
                action = black.action(observation.black, observation.white)
            else:
                action = white.action(observation.white, observation.black)
            observation, info = env.step(action)
        ng_win = None
        if env.winner == Winner.black:
            ng_win = 0 if best_is_black else 1
        elif env.winner == Winner.white:
            ng_win = 1 if best_is_black else 0
        return ng_win, best_is_black, observation.number_of_black_and_white

    def load_best_model(self):
        model = ReversiModel(self.config)
        if not load_best_model_weight(model):
            raise RuntimeError("Best model can not loaded!")
        return model

    def load_next_generation_model(self, poll_seconds=60):
        rc = self.config.resource
        while True:
            dirs = get_next_generation_model_dirs(rc)
            if dirs:
                break
            logger.info("There is no next generation model to evaluate")
            sleep(poll_seconds)
        model_dir = dirs[-1] if self.config.eval.evaluate_latest_first else dirs[0]
        config_path = os.path.join(model_dir, rc.next_generation_model_config_filename)
        weight_path = os.path.join(model_dir, rc.next_generation_model_weight_filename)
        model = ReversiModel(self.config)
        model.load(config_path, weight_path)
        return model, model_dir

    def remove_model(self, model_dir):
        rc = self.config.resource
        os.remove(os.path.join(model_dir, rc.next_generation_model_config_filename))
        os.remove(os.path.join(model_dir, rc.next_generation_model_weight_filename))
        os.rmdir(model_dir)
