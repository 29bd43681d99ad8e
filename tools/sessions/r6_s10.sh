#!/bin/bash
# Round 6, GPU session 10: slices (streams) of the as-shipped lock-step leg on the final build: (budget, waves, fused, parts, every, continuous)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r6_s10; mkdir -p $OUT
cd $ROOT
timeout 600 python tools/sessions/quick_solver_ab.py "0,0,0,1;0,0,0,2;0,0,0,3;0,0,0,4;256,0,0,1;256,0,0,2;0,0,0,6;0,0,0,8" > $OUT/parts_ab.jsonl 2> $OUT/ab.err
python - <<PY
import json
for line in open("$OUT/parts_ab.jsonl"):
    try: d = json.loads(line)
    except Exception: continue
    sp = d.get("solver_pool") or {}
    print({k: d.get(k) for k in ("budget", "parts", "every")}, "sims/s %.2f M" % (d["sims_per_s"] / 1e6), "steps", d.get("steps"), "ms/step %.3f" % d.get("ms_per_step", 0), "rounds/answer", sp.get("pool_rounds_per_answer"), "util", sp.get("lane_utilisation"))
PY
tail -2 $OUT/ab.err | cut -c1-200
