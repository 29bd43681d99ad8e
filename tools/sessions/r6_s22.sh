#!/bin/bash
# Round 6, GPU session 22: the solver's move function finishes a position with ONE empty square itself (RAZ_SOLVER_INLINE_LAST) instead of
# handing it back as a node that costs the worker wave an iteration - parity (the solver GPU tests), the timeline, A/B of the legs.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r6_s22; mkdir -p $OUT
cd $ROOT
RAZ_LIB_PATH=$ROOT/build/variants/libraz_inl2.so timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_engine_par_gpu.py tests/test_zz_fused_gpu.py tests/test_continuous_gpu.py -q -m gpu -x -k "solver or solve or shipped" > $OUT/pytest_solver_inl2.log 2>&1; echo "pytest solver (inl2) rc=$?"; tail -2 $OUT/pytest_solver_inl2.log
RAZ_LIB_PATH=$ROOT/build/variants/libraz_inl2.so RAZ_TIMELINE_TIMED=0 timeout 300 python tools/solver_timeline.py > $OUT/timeline_inl2.json 2>> $OUT/err.log
LEGS=config1_mini_yml_as_shipped,config1_mini_yml_as_shipped_two_kernel_pipeline,config1_mini_yml_as_shipped_continuous_batching,ch5_yml_as_shipped
for round in 1 2; do
for v in inl1 inl2; do
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
