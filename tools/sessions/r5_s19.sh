#!/bin/bash
# Round 5, GPU session 19: tree launches per round of the solver pool (and the round's budget) on mini.yml as shipped with continuous
# batching - the worker's mode - where session 18 measured 22.7 M (every step waits) -> 30.0 M sims/s (the round beside four steps).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r5_s19; mkdir -p $OUT
cd $ROOT
timeout 420 python tools/sessions/quick_solver_ab.py "0,0,0,0,2,1;0,0,0,0,3,1;0,0,0,0,6,1;64,0,0,0,4,1;256,0,0,0,4,1;0,0,0,2,3,1;64,0,0,0,2,1" > $OUT/ab.jsonl 2> $OUT/ab.err
echo "ab rc=$?"
python - <<PY
import json
for line in open("$OUT/ab.jsonl"):
    try: d = json.loads(line)
    except Exception: continue
    print({k: d.get(k) for k in ("budget", "fused", "parts", "every", "continuous")}, "sims/s %.2f M" % (d["sims_per_s"] / 1e6), "steps", d.get("steps"), "ms/step %.3f" % d.get("ms_per_step", 0))
PY
tail -3 $OUT/ab.err | cut -c1-300
