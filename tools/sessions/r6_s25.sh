#!/bin/bash
# Round 6, GPU session 25: TWO worker waves of the solver pool per SIMD - the lanes' frames in LDS cut to 8 levels (16 KB per wave instead
# of 28 KB: what use_solver_turn 50 needs), 2048 worker waves against 1024.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r6_s25; mkdir -p $OUT
cd $ROOT
LEGS=config1_mini_yml_as_shipped_two_kernel_pipeline,config1_mini_yml_as_shipped_continuous_batching
run() {
  name=$1; lib=$2; shift 2
  env RAZ_LIB_PATH=$ROOT/build/variants/libraz_$lib.so "$@" timeout 600 python bench.py --no-cpu-baseline --no-whole-games --steps 3 --warmup 1 --no-spotcheck --legs $LEGS --full-out $OUT/$name.json > /dev/null 2> $OUT/$name.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/$name.json"))
    print("$name", {k.replace("config1_mini_yml_", ""): (round(d[k]["value"] / 1e6, 3), d[k].get("steps"), round((d[k].get("solver_pool") or {}).get("pool_rounds_per_answer") or 0, 2), (d[k].get("solver_pool") or {}).get("ticks", {}).get("ticks_per_wave_iteration"), d[k].get("parity") or d[k].get("parity_check")) if isinstance(d.get(k), dict) and d[k].get("value") else str(d.get(k))[:120] for k in "$LEGS".split(",")})
except Exception as e:
    print("$name", "no result", e)
PY
}
run base_1024 base X=1
run lds8_1024 lds8 X=1
run lds8_2048 lds8 RAZ_BENCH_SOLVER_WAVES=2048
run lds8_2048_b64 lds8 RAZ_BENCH_SOLVER_WAVES=2048 RAZ_BENCH_SOLVER_BUDGET=64
run lds8_1536 lds8 RAZ_BENCH_SOLVER_WAVES=1536
run lds8_2048_again lds8 RAZ_BENCH_SOLVER_WAVES=2048
