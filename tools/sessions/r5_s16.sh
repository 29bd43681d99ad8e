#!/bin/bash
# Round 5, GPU session 16: the fused tree + net kernels without VGPR spills.  (1) A/B on one box of configs[1] on the fused kernel: the
# previous commit's library, this build, and this build at 64 / 256 simulation steps per launch (libraries built beforehand under build/variants,
# chosen with RAZ_LIB_PATH); (2) the GPU parity tests of the fused kernels and the solver pool on this build; (3) the counter passes the
# bench line's traffic figures rest on, retaken on these sources (profiles/r5_pmc carries the sources' sha256), + the fused form's own passes.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r5_s16; mkdir -p $OUT
cd $ROOT
LEGS=config1_4096x200_mini,config1_mini_yml_parallel_search_num_4
for v in head cur it64 it256 cur2 head2; do
  case $v in cur*) unset RAZ_LIB_PATH;; head*) export RAZ_LIB_PATH=$ROOT/build/variants/libraz_head.so;; *) export RAZ_LIB_PATH=$ROOT/build/variants/libraz_$v.so;; esac
  timeout 300 python bench.py --no-cpu-baseline --no-whole-games --steps 3 --warmup 1 --no-spotcheck --legs $LEGS --full-out $OUT/ab_$v.json > /dev/null 2> $OUT/ab_$v.err
  python - <<PY
import json
d = json.load(open("$OUT/ab_$v.json"))
print("$v", {k: (round(d[k]["value"] / 1e6, 2) if isinstance(d.get(k), dict) and d[k].get("value") else d.get(k)) for k in "$LEGS".split(",")})
PY
done
unset RAZ_LIB_PATH
timeout 500 python -m pytest tests/test_zz_fused_gpu.py tests/test_engine_gpu.py -q -m gpu -x -k "fused or solver_pool or suspended or with_solver_batch" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
PROF_TIMEOUT=300 bash tools/run_profiles.sh headline 20 r5_s16/prof_headline "stats 3 4" > $OUT/prof_headline.log 2>&1; tail -2 $OUT/prof_headline.log
PROF_TIMEOUT=200 bash tools/run_profiles.sh headline 600 r5_s16/prof_config1 "3 4" --net mini --games 4096 --sims 200 > $OUT/prof_config1.log 2>&1; tail -2 $OUT/prof_config1.log
PROF_TIMEOUT=200 bash tools/run_profiles.sh headline 600 r5_s16/prof_config1_fused "stats 3 4" --net mini --games 4096 --sims 200 --fused > $OUT/prof_config1_fused.log 2>&1; tail -2 $OUT/prof_config1_fused.log
find "$OUT" -name "*_kernel_trace.csv" -delete; find "$OUT" -name "*_counter_collection.csv" -delete
