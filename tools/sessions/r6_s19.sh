#!/bin/bash
# Round 6, GPU session 19: with the draw gate in (session 18), the pool's knobs again on the mini.yml-as-shipped legs: iterations per round
# (budget), worker waves, rounds beside n tree launches (lock-step two-kernel and continuous batching).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r6_s19; mkdir -p $OUT
cd $ROOT
export RAZ_LIB_PATH=$ROOT/build/variants/libraz_gate1.so
run() {  # name, legs, env...
  name=$1; legs=$2; shift 2
  env "$@" timeout 400 python bench.py --no-cpu-baseline --no-whole-games --steps 3 --warmup 1 --no-spotcheck --legs $legs --full-out $OUT/$name.json > /dev/null 2> $OUT/$name.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/$name.json"))
    print("$name", {k.replace("config1_mini_yml_", ""): (round(d[k]["value"] / 1e6, 2), d[k].get("steps"), round((d[k].get("solver_pool") or {}).get("pool_rounds_per_answer") or 0, 2)) if isinstance(d.get(k), dict) and d[k].get("value") else str(d.get(k))[:80] for k in "$legs".split(",")})
except Exception as e:
    print("$name", "no result", e)
PY
}
L2=config1_mini_yml_as_shipped_two_kernel_pipeline
LC=config1_mini_yml_as_shipped_continuous_batching
run base $L2,$LC X=1
for b in 64 192 256 384; do run budget_$b $L2,$LC RAZ_BENCH_SOLVER_BUDGET=$b; done
for w in 768 1280 1536; do run waves_$w $L2,$LC RAZ_BENCH_SOLVER_WAVES=$w; done
for n in 2 3; do run every_$n $L2,$LC RAZ_TUNING=1 RAZ_SOLVER_POOL_EVERY=$n RAZ_BENCH_POOL_EVERY=$n; done
run base_again $L2,$LC X=1
