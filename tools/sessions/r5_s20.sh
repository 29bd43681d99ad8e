#!/bin/bash
# Round 5, GPU session 20 (final sources): the counter passes the bench line's traffic figures rest on (profiles/r5_pmc carries the
# kernel sources' sha256), the GPU parity tests of the fused kernels / solver pool, and the configs[1] legs whose kernels changed
# (k_tree_par_net without spills, 256 steps per fused launch, the solver pool's round beside the tree launches under continuous batching).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r5_s20; mkdir -p $OUT
cd $ROOT
PROF_TIMEOUT=200 bash tools/run_profiles.sh headline 20 r5_s20/prof_headline "stats 3 4" > $OUT/prof_headline.log 2>&1; tail -2 $OUT/prof_headline.log
timeout 200 python -m pytest tests/test_zz_fused_gpu.py tests/test_engine_gpu.py -q -m gpu -x -k "fused or solver_pool or suspended or with_solver_batch" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest.log
LEGS=config1_4096x200_mini,config1_mini_yml_parallel_search_num_4,config1_mini_yml_as_shipped,config1_mini_yml_as_shipped_two_kernel_pipeline,config1_mini_yml_as_shipped_continuous_batching
timeout 400 python bench.py --no-cpu-baseline --no-whole-games --steps 3 --warmup 1 --legs $LEGS --full-out $OUT/legs_full.json > $OUT/legs_line.json 2> $OUT/legs.err; echo "legs rc=$?"
python - <<PY
import json
d = json.load(open("$OUT/legs_full.json"))
for k in "$LEGS".split(","):
    v = d.get(k) or {}
    print(k, round((v.get("value") or 0) / 1e6, 2), "M sims/s", "games/h", v.get("games_per_hour"), "spot", (v.get("parity_spotcheck") or {}).get("result") if isinstance(v.get("parity_spotcheck"), dict) else v.get("parity_spotcheck"), str(v.get("error"))[:200])
PY
PROF_TIMEOUT=150 bash tools/run_profiles.sh headline 600 r5_s20/prof_config1 "3 4" --net mini --games 4096 --sims 200 > $OUT/prof_config1.log 2>&1; tail -2 $OUT/prof_config1.log
find "$OUT" -name "*_kernel_trace.csv" -delete; find "$OUT" -name "*_counter_collection.csv" -delete
