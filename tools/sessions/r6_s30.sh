#!/bin/bash
# Round 6, GPU session 30: hardware queues - the runtime maps a process's streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues; 3 slices
# + 3 pool streams + the caller's stream are more than that.  mini.yml as shipped and configs[1] on the final build with 4 / 8 queues.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r6_s30; mkdir -p $OUT
cd $ROOT
LEGS=config1_mini_yml_as_shipped_two_kernel_pipeline,config1_mini_yml_as_shipped_continuous_batching,config1_two_kernel_pipeline,config1_continuous_batching
run() {
  name=$1; shift
  env "$@" timeout 600 python bench.py --no-cpu-baseline --no-whole-games --steps 3 --warmup 1 --no-spotcheck --legs $LEGS --full-out $OUT/$name.json > /dev/null 2> $OUT/$name.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/$name.json"))
    print("$name", "headline %.1f k" % (d["value"] / 1e3), {k.replace("config1_", "").replace("mini_yml_", ""): (round(d[k]["value"] / 1e6, 3), d[k].get("steps")) if isinstance(d.get(k), dict) and d[k].get("value") else str(d.get(k))[:160] for k in "$LEGS".split(",")})
except Exception as e:
    print("$name", "no result", e)
PY
}
run q4 X=1
run q8 GPU_MAX_HW_QUEUES=8
run q8_parts4 GPU_MAX_HW_QUEUES=8 RAZ_BENCH_PARTS=4
run q2 GPU_MAX_HW_QUEUES=2
run q8_again GPU_MAX_HW_QUEUES=8
