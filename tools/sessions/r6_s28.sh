#!/bin/bash
# Round 6, GPU session 28: k_solve_scan's pass BEFORE a round no longer walks the task trees of solves that are already running (nothing
# has run since the pass after the last round) - parity (the solver GPU tests), A/B of the as-shipped legs.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r6_s28; mkdir -p $OUT
cd $ROOT
RAZ_LIB_PATH=$ROOT/build/variants/libraz_scan.so timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_engine_par_gpu.py tests/test_zz_fused_gpu.py tests/test_continuous_gpu.py -q -m gpu -x -k "solver or solve or shipped" > $OUT/pytest_solver_scan.log 2>&1; echo "pytest solver (scan) rc=$?"; tail -2 $OUT/pytest_solver_scan.log
LEGS=config1_mini_yml_as_shipped,config1_mini_yml_as_shipped_two_kernel_pipeline,config1_mini_yml_as_shipped_continuous_batching,ch5_yml_as_shipped
for round in 1 2; do
for v in base scan; do
  export RAZ_LIB_PATH=$ROOT/build/variants/libraz_$v.so
  timeout 600 python bench.py --no-cpu-baseline --no-whole-games --steps 3 --warmup 1 --no-spotcheck --legs $LEGS --full-out $OUT/ab_${v}_$round.json > /dev/null 2> $OUT/ab_${v}_$round.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/ab_${v}_$round.json"))
    print("$v", {k.replace("config1_mini_yml_", ""): (round(d[k]["value"] / 1e6, 3), d[k].get("steps"), round((d[k].get("solver_pool") or {}).get("pool_rounds_per_answer") or 0, 2)) if isinstance(d.get(k), dict) and d[k].get("value") else str(d.get(k))[:80] for k in "$LEGS".split(",")})
except Exception as e:
    print("$v", "no result", e)
PY
done
done
