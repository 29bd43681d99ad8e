#!/bin/bash
# Round 5, GPU session 9: the as-shipped legs with their parity checks, the as-shipped steady state, the worker end to end.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r5_s9; mkdir -p $OUT
cd $ROOT
timeout 1200 python bench.py --no-cpu-baseline --no-whole-games --steps 5 --warmup 2 --no-spotcheck --legs ${1:-config1_mini_yml_as_shipped,config1_mini_yml_as_shipped_two_kernel_pipeline,config1_mini_yml_as_shipped_continuous_batching,worker_end_to_end_config1} --full-out $OUT/legs_full.json > $OUT/legs_line.json 2> $OUT/legs.err
echo "legs rc=$?"; tail -5 $OUT/legs.err
python - <<PY
import json
d = json.load(open("$OUT/legs_full.json"))
for k, v in d.items():
    if isinstance(v, dict) and ("as_shipped" in k or "worker" in k):
        print(k, json.dumps({x: v.get(x) for x in v if x not in ("workload", "roofline", "solver_pool", "seconds")})[:1500])
        if "solver_pool" in v: print("   pool", json.dumps(v["solver_pool"])[:700])
PY
