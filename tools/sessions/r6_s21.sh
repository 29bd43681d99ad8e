#!/bin/bash
# Round 6, GPU session 21: the worker loop's compile-time knobs again on the build with gated task draws (slow phase every 8 / 16 / 32
# iterations, lane memo from 6 / 7 / 8 empties, at most 1 / 2 / 3 returns per iteration) - mini.yml as shipped, lock-step and continuous.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r6_s21; mkdir -p $OUT
cd $ROOT
LEGS=config1_mini_yml_as_shipped_two_kernel_pipeline,config1_mini_yml_as_shipped_continuous_batching
for v in base slow8 slow32 memo6 memo8 ret1 ret3 base; do
  export RAZ_LIB_PATH=$ROOT/build/variants/libraz_$v.so
  timeout 400 python bench.py --no-cpu-baseline --no-whole-games --steps 3 --warmup 1 --no-spotcheck --legs $LEGS --full-out $OUT/ab_${v}.json > /dev/null 2> $OUT/ab_${v}.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/ab_${v}.json"))
    print("$v", {k.replace("config1_mini_yml_", ""): (round(d[k]["value"] / 1e6, 2), d[k].get("steps"), round((d[k].get("solver_pool") or {}).get("pool_rounds_per_answer") or 0, 2), round((d[k].get("solver_pool") or {}).get("lane_utilisation") or 0, 3)) if isinstance(d.get(k), dict) and d[k].get("value") else str(d.get(k))[:80] for k in "$LEGS".split(",")})
except Exception as e:
    print("$v", "no result", e)
PY
done
