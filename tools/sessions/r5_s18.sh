#!/bin/bash
# Round 5, GPU session 18: the solver pool's round on its own stream beside the tree kernels (raz_engine_config.reserved bits 24-27 =
# tree launches per round).  (1) the solver parity tests on the GPU as built (every step waits for the round: checks the answer now
# carried by the header's state word) and with RAZ_SOLVER_POOL_EVERY=4 (the round overlaps three tree launches); (2) A/B on mini.yml
# as shipped: lock-step whole games two-kernel / fused, and continuous batching.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r5_s18; mkdir -p $OUT
cd $ROOT
SEL="fused or solver_pool or suspended or with_solver_batch"
timeout 300 python -m pytest tests/test_zz_fused_gpu.py tests/test_engine_gpu.py -q -m gpu -x -k "$SEL" > $OUT/pytest_every1.log 2>&1; echo "pytest every=1 rc=$?"; tail -2 $OUT/pytest_every1.log
RAZ_SOLVER_POOL_EVERY=4 timeout 300 python -m pytest tests/test_zz_fused_gpu.py tests/test_engine_gpu.py -q -m gpu -x -k "$SEL" > $OUT/pytest_every4.log 2>&1; echo "pytest every=4 rc=$?"; tail -2 $OUT/pytest_every4.log
timeout 420 python tools/sessions/quick_solver_ab.py "0,0,0,0,1;0,0,0,0,2;0,0,0,0,4;0,0,0,0,8;0,0,1,0,1;0,0,1,0,2;0,0,1,0,4;64,0,0,0,4;64,0,0,0,8;0,0,0,2,4;0,0,0,0,1,1;0,0,0,0,4,1" > $OUT/ab.jsonl 2> $OUT/ab.err
echo "ab rc=$?"
python - <<PY
import json
for line in open("$OUT/ab.jsonl"):
    try: d = json.loads(line)
    except Exception: continue
    sp = d.get("solver_pool") or {}
    print({k: d.get(k) for k in ("budget", "fused", "parts", "every", "continuous")}, "sims/s %.2f M" % (d["sims_per_s"] / 1e6), "steps", d.get("steps"),
          "ms/step %.3f" % d.get("ms_per_step", 0), "rounds/answer", sp.get("rounds_per_answer"), "util", sp.get("lane_utilisation"))
PY
tail -3 $OUT/ab.err | cut -c1-300
