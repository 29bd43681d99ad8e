#!/bin/bash
# Round 6, GPU session 29: slices of the batch (streams) on the final build, mini.yml as shipped.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r6_s29; mkdir -p $OUT
cd $ROOT
LEGS=config1_mini_yml_as_shipped_two_kernel_pipeline,config1_mini_yml_as_shipped_continuous_batching
for n in 0 2 4 0; do
  RAZ_BENCH_PARTS=$n timeout 600 python bench.py --no-cpu-baseline --no-whole-games --steps 3 --warmup 1 --no-spotcheck --legs $LEGS --full-out $OUT/parts_$n.json > /dev/null 2> $OUT/parts_$n.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/parts_$n.json"))
    print("parts $n", {k.replace("config1_mini_yml_", ""): (round(d[k]["value"] / 1e6, 3), d[k].get("steps")) if isinstance(d.get(k), dict) and d[k].get("value") else str(d.get(k))[:160] for k in "$LEGS".split(",")})
except Exception as e:
    print("parts $n", "no result", e)
PY
done
