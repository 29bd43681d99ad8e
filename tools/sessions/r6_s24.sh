#!/bin/bash
# Round 6, GPU session 24: inline-last level 2 (and 1) with the library's default of 96 / 80 iterations per round.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r6_s24; mkdir -p $OUT
cd $ROOT
LEGS=config1_mini_yml_as_shipped,config1_mini_yml_as_shipped_two_kernel_pipeline,config1_mini_yml_as_shipped_continuous_batching
for v in inl2b96 inl2b80 inl1b96 inl2b96; do
  export RAZ_LIB_PATH=$ROOT/build/variants/libraz_$v.so
  timeout 600 python bench.py --no-cpu-baseline --no-whole-games --steps 3 --warmup 1 --no-spotcheck --legs $LEGS --full-out $OUT/ab_${v}.json > /dev/null 2> $OUT/ab_${v}.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/ab_${v}.json"))
    print("$v", {k.replace("config1_mini_yml_", ""): (round(d[k]["value"] / 1e6, 3), d[k].get("steps"), round((d[k].get("solver_pool") or {}).get("pool_rounds_per_answer") or 0, 2)) if isinstance(d.get(k), dict) and d[k].get("value") else str(d.get(k))[:80] for k in "$LEGS".split(",")})
except Exception as e:
    print("$v", "no result", e)
PY
done
