"""Is raz_engine_step host-launch-bound?  Times the enqueue (host) and the completion (GPU) of n steps."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from reversi_alpha_zero_amd.config import Config, load_config
from reversi_alpha_zero_amd.agent.model import ReversiNet
from reversi_alpha_zero_amd.engine import DeviceNet, SelfPlayEngine

cfg = Config()
cfg.play.thinking_loop = 1
cfg.play.use_solver_turn = 0
cfg.play.use_solver_turn_in_simulation = 0
cfg.play.share_mtcs_info_in_self_play = True
cfg.play.parallel_search_num = 1
net = DeviceNet(ReversiNet(16, 1, 16).keras_init_(0).to_blob(), "cuda:0")
for parts in (1, 2, 3, 4, 5, 6):
    eng = SelfPlayEngine(cfg, net, 4096, seed=0, sims_hint=200, parts=parts)
    eng.start(0, 200)
    eng.step(500)
    torch.cuda.synchronize()
    n = 2000
    t0 = time.perf_counter()
    eng.step(n)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"parts={parts}: host enqueue {1e6*(t1-t0)/n:.1f} us/step ({1e6*(t1-t0)/n/(2*parts):.2f} us/launch), "
          f"total {1e6*(t2-t0)/n:.1f} us/step")
    del eng
