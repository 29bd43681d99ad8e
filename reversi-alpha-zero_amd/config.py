"""Hyper-parameter tree with the reference's names and default values (reversi_zero/config.py:15-193)
for the sections the self-play path reads: play, play_data, model, resource (+ opts).  The trainer /
eval / GUI sections of the reference are outside the hot path and are not restated.

`load_config(yml_path)` overlays a reference YAML file (config/*.yml) the way
manager.py:41-45 + moke_config.create_config do: recursive attribute assignment, unknown sections
are kept as plain attributes so a reference yml loads unchanged.
"""
import os


class _Section:
    def update(self, d):
        for k, v in d.items():
            cur = getattr(self, k, None)
            if isinstance(v, dict) and isinstance(cur, _Section):
                cur.update(v)
            else:
                setattr(self, k, v)
        return self

    def as_dict(self):
        return {k: (v.as_dict() if isinstance(v, _Section) else v) for k, v in vars(self).items()}


class Options(_Section):
    def __init__(self):
        self.new = False


class ResourceConfig(_Section):
    """Paths (config.py:33-58); DATA_DIR / MODEL_DIR / PROJECT_DIR env overrides honoured."""

    def __init__(self):
        here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        self.project_dir = os.environ.get("PROJECT_DIR", here)
        self.data_dir = os.environ.get("DATA_DIR", os.path.join(self.project_dir, "data"))
        self.model_dir = os.environ.get("MODEL_DIR", os.path.join(self.data_dir, "model"))
        self.model_best_config_path = os.path.join(self.model_dir, "model_best_config.json")
        self.model_best_weight_path = os.path.join(self.model_dir, "model_best_weight.h5")
        self.next_generation_model_dir = os.path.join(self.model_dir, "next_generation")
        self.next_generation_model_dirname_tmpl = "model_%s"
        self.next_generation_model_config_filename = "model_config.json"
        self.next_generation_model_weight_filename = "model_weight.h5"
        self.play_data_dir = os.path.join(self.data_dir, "play_data")
        self.play_data_filename_tmpl = "play_%s.json"
        self.self_play_ggf_data_dir = os.path.join(self.data_dir, "self_play-ggf")
        self.ggf_filename_tmpl = "self_play-%s.ggf"
        self.log_dir = os.path.join(self.project_dir, "logs")
        self.main_log_path = os.path.join(self.log_dir, "main.log")
        self.force_simulation_num_file = os.path.join(self.data_dir, ".force-sim")
        self.self_play_game_idx_file = os.path.join(self.data_dir, ".self-play-game-idx")

    def create_directories(self):
        for d in (self.project_dir, self.data_dir, self.model_dir, self.play_data_dir, self.log_dir,
                  self.next_generation_model_dir, self.self_play_ggf_data_dir):
            os.makedirs(d, exist_ok=True)


class PlayDataConfig(_Section):
    def __init__(self):  # config.py:116-125
        self.multi_process_num = 16
        self.nb_game_in_file = 2
        self.max_file_num = 800
        self.save_policy_of_tau_1 = True
        self.enable_ggf_data = True
        self.nb_game_in_ggf_file = 100
        self.drop_draw_game_rate = 0


class PlayConfig(_Section):
    def __init__(self):  # config.py:128-166
        self.simulation_num_per_move = 200
        self.share_mtcs_info_in_self_play = True
        self.reset_mtcs_info_per_game = 1
        self.thinking_loop = 10
        self.required_visit_to_decide_action = 400
        self.start_rethinking_turn = 8
        self.c_puct = 1
        self.noise_eps = 0.25
        self.dirichlet_alpha = 0.5
        self.change_tau_turn = 4
        self.virtual_loss = 3
        self.prediction_queue_size = 16
        self.parallel_search_num = 8
        self.prediction_worker_sleep_sec = 0.0001
        self.wait_for_expanding_sleep_sec = 0.00001
        self.resign_threshold = -0.9
        self.allowed_resign_turn = 20
        self.disable_resignation_rate = 0.1
        self.false_positive_threshold = 0.05
        self.resign_threshold_delta = 0.01
        self.policy_decay_turn = 60  # inert in the reference
        self.policy_decay_power = 3
        self.use_solver_turn = 50
        self.use_solver_turn_in_simulation = 50
        self.schedule_of_simulation_num_per_move = [(0, 8), (300, 50), (2000, 200)]
        self.use_newest_next_generation_model = True


class EvaluateConfig(_Section):
    """config.py:101-110.  play_config.parallel_search_num stays at the PlayConfig default (8), as in the
    reference; the engine plays it on the deterministic raz-sched-v1 schedule."""

    def __init__(self):
        self.game_num = 200
        self.replace_rate = 0.55
        self.play_config = PlayConfig()
        self.play_config.simulation_num_per_move = 400
        self.play_config.thinking_loop = 1
        self.play_config.change_tau_turn = 0
        self.play_config.noise_eps = 0
        self.play_config.disable_resignation_rate = 0
        self.evaluate_latest_first = True


class ModelConfig(_Section):
    def __init__(self):  # config.py:187-193
        self.cnn_filter_num = 256
        self.cnn_filter_size = 3
        self.res_layer_num = 10
        self.l2_reg = 1e-4
        self.value_fc_size = 256


class Config(_Section):
    def __init__(self):
        self.type = "default"
        self.opts = Options()
        self.resource = ResourceConfig()
        self.model = ModelConfig()
        self.play = PlayConfig()
        self.play_data = PlayDataConfig()
        self.eval = EvaluateConfig()


def load_config(yml_path=None, overrides=None):
    cfg = Config()
    if yml_path:
        import yaml
        with open(yml_path, "rt") as f:
            cfg.update(yaml.safe_load(f) or {})
    if overrides:
        cfg.update(overrides)
    return cfg
