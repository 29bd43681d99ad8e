#!/bin/bash
# Round 6, GPU session 31: 6 / 12 / 16 hardware queues on the continuous-batching legs.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r6_s31; mkdir -p $OUT
cd $ROOT
LEGS=config1_mini_yml_as_shipped_continuous_batching,config1_mini_yml_as_shipped_two_kernel_pipeline
for q in 6 12 16 8; do
  GPU_MAX_HW_QUEUES=$q timeout 600 python bench.py --no-cpu-baseline --no-whole-games --steps 3 --warmup 1 --no-spotcheck --legs $LEGS --full-out $OUT/q$q.json > /dev/null 2> $OUT/q$q.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/q$q.json"))
    print("queues $q", "headline %.1f k" % (d["value"] / 1e3), {k.replace("config1_", "").replace("mini_yml_", ""): (round(d[k]["value"] / 1e6, 3), d[k].get("steps")) if isinstance(d.get(k), dict) and d[k].get("value") else str(d.get(k))[:160] for k in "$LEGS".split(",")})
except Exception as e:
    print("queues $q", "no result", e)
PY
done
