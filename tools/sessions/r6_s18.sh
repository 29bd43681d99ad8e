#!/bin/bash
# Round 6, GPU session 18: the solver pool's task draws gated by the tasks a solve has left (RAZ_SOLVER_DRAW_GATE) - the timeline of the
# mini.yml-as-shipped leg (tools/solver_timeline.py) showed rounds of 1 ms at the start and at the end of a batch, when 65 000 idle lanes
# draw with atomics from a handful of listed solves.  Parity (the solver GPU tests on the new library), the timeline, then A/B of the legs.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r6_s18; mkdir -p $OUT
cd $ROOT
RAZ_LIB_PATH=$ROOT/build/variants/libraz_gate1.so timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_engine_par_gpu.py tests/test_zz_fused_gpu.py tests/test_continuous_gpu.py -q -m gpu -x -k "solver or solve or shipped" > $OUT/pytest_solver_gate1.log 2>&1; echo "pytest solver (gate1) rc=$?"; tail -2 $OUT/pytest_solver_gate1.log
for v in gate0 gate1; do
  RAZ_LIB_PATH=$ROOT/build/variants/libraz_$v.so RAZ_TIMELINE_TIMED=0 timeout 300 python tools/solver_timeline.py > $OUT/timeline_$v.json 2>> $OUT/err.log
done
LEGS=config1_mini_yml_as_shipped,config1_mini_yml_as_shipped_two_kernel_pipeline,config1_mini_yml_as_shipped_continuous_batching
for round in 1 2; do
for v in gate0 gate1; do
  export RAZ_LIB_PATH=$ROOT/build/variants/libraz_$v.so
  timeout 600 python bench.py --no-cpu-baseline --no-whole-games --steps 3 --warmup 1 --no-spotcheck --legs $LEGS --full-out $OUT/ab_${v}_$round.json > /dev/null 2> $OUT/ab_${v}_$round.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/ab_${v}_$round.json"))
    print("$v", {k.replace("config1_mini_yml_", ""): (round(d[k]["value"] / 1e6, 2), d[k].get("steps"), (d[k].get("solver_pool") or {}).get("pool_rounds_per_answer")) if isinstance(d.get(k), dict) and d[k].get("value") else str(d.get(k))[:80] for k in "$LEGS".split(",")})
except Exception as e:
    print("$v", "no result", e)
PY
done
done
