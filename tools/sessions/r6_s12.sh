#!/bin/bash
# Round 6, GPU session 12: lock-step as-shipped leg with the pool's round BESIDE the next tree launches (every = 2, 3) at smaller pools - does the
# round overlap once its waves leave LDS for the tree kernels?  (budget, waves, fused, parts, every, continuous)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r6_s12; mkdir -p $OUT
cd $ROOT
timeout 700 python tools/sessions/quick_solver_ab.py "0,1024,0,0,1;0,1024,0,0,2;0,512,0,0,1;0,512,0,0,2;0,512,0,0,3;0,256,0,0,2;0,256,0,0,3;0,768,0,0,2;64,512,0,0,2;256,512,0,0,2" > $OUT/overlap_ab.jsonl 2> $OUT/ab.err
python - <<PY
import json
for line in open("$OUT/overlap_ab.jsonl"):
    try: d = json.loads(line)
    except Exception: continue
    sp = d.get("solver_pool") or {}
    print({k: d.get(k) for k in ("budget", "waves", "every")}, "sims/s %.2f M" % (d["sims_per_s"] / 1e6), "steps", d.get("steps"), "ms/step %.3f" % d.get("ms_per_step", 0), "rounds/answer", sp.get("pool_rounds_per_answer"))
PY
tail -2 $OUT/ab.err | cut -c1-200
