#!/bin/bash
# Round 6, GPU session 9: does the as-shipped leg scale with the solver pool's size on the final build?  (budget, waves, fused, parts, every, continuous)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r6_s9; mkdir -p $OUT
cd $ROOT
timeout 600 python tools/sessions/quick_solver_ab.py "0,512,0;0,768,0;0,1024,0;0,1280,0;0,512,0,0,3,1;0,1024,0,0,3,1;0,1280,0,0,3,1;64,1280,0;256,1280,0" > $OUT/waves_ab.jsonl 2> $OUT/ab.err
python - <<PY
import json
for line in open("$OUT/waves_ab.jsonl"):
    try: d = json.loads(line)
    except Exception: continue
    sp = d.get("solver_pool") or {}
    print({k: d.get(k) for k in ("budget", "waves", "every", "continuous")}, "sims/s %.2f M" % (d["sims_per_s"] / 1e6), "steps", d.get("steps"), "ms/step %.3f" % d.get("ms_per_step", 0), "rounds/answer", sp.get("pool_rounds_per_answer"), "util", sp.get("lane_utilisation"), "waves", sp.get("worker_waves"))
PY
tail -2 $OUT/ab.err | cut -c1-200
