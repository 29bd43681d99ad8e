"""GPU A/B of the solver pool's knobs on mini.yml as shipped (configs[1] batch): one process, several (budget, waves, fused) settings.
usage: python tools/sessions/quick_solver_ab.py "budget,waves,fused[,parts[,every[,continuous]]];..."   (every = tree launches per round of
the pool, raz_engine_config.reserved bits 24-27; continuous = 1: the continuous-batching leg - 12 288 game ids on 4096 slots - instead) """
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402
import __graft_entry__ as g  # noqa: E402

g.build()
dev = torch.device("cuda:0")
args = types.SimpleNamespace(no_spotcheck=True)
for spec in (sys.argv[1] if len(sys.argv) > 1 else "0,0,0").split(";"):
    b, w, f, parts, every, cont = (list(int(x) for x in spec.split(",")) + [0, 0, 0])[:6]
    os.environ["RAZ_BENCH_SOLVER_BUDGET"], os.environ["RAZ_BENCH_SOLVER_WAVES"], os.environ["RAZ_BENCH_PARTS"] = str(b), str(w), str(parts)
    os.environ["RAZ_BENCH_POOL_EVERY"] = str(every)
    knobs = {"budget": b, "waves": w, "fused": f, "parts": parts, "every": every, "continuous": cont}
    if cont:
        out = bench.continuous_leg(dev, args, shipped=True)
        print(json.dumps(dict(knobs, sims_per_s=out.get("value") or out.get("sims_per_s"), games_per_hour=out.get("games_per_hour"),
                              **{k: out[k] for k in ("steps", "ms_per_step", "seconds") if k in out})), flush=True)
        continue
    out = bench.config1_leg(dev, args, 4, fused=bool(f), shipped=True)[0]
    print(json.dumps(dict(knobs, sims_per_s=out["value"], steps=out["steps"], ms_per_step=out["ms_per_step"],
                          k_tree_avg_ms=out.get("k_tree_avg_ms"), solver_pool=out["solver_pool"])), flush=True)
