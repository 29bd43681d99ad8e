#!/bin/bash
# Round 6, GPU session 26: the pool's round beside n tree launches on the final build (last square in the move function, 96 iterations per
# round) - continuous batching and lock-step, mini.yml as shipped.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r6_s26; mkdir -p $OUT
cd $ROOT
LEGS=config1_mini_yml_as_shipped_two_kernel_pipeline,config1_mini_yml_as_shipped_continuous_batching
for n in 1 2 3 4 1; do
  RAZ_BENCH_POOL_EVERY=$n timeout 600 python bench.py --no-cpu-baseline --no-whole-games --steps 3 --warmup 1 --no-spotcheck --legs $LEGS --full-out $OUT/every_$n.json > /dev/null 2> $OUT/every_$n.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/every_$n.json"))
    print("every $n", {k.replace("config1_mini_yml_", ""): (round(d[k]["value"] / 1e6, 3), d[k].get("steps")) if isinstance(d.get(k), dict) and d[k].get("value") else str(d.get(k))[:120] for k in "$LEGS".split(",")})
except Exception as e:
    print("every $n", "no result", e)
PY
done
