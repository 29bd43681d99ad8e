#!/bin/bash
# Round 5, final GPU session: the whole GPU suite, the default bench run, and every profile the line's figures rest on, all on the final sources.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r5_s14; mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
timeout 900 python bench.py --full-out $OUT/bench_full.json > $OUT/bench_line.json 2> $OUT/bench.err
echo "bench rc=$? lines=$(wc -l < $OUT/bench_line.json) bytes=$(wc -c < $OUT/bench_line.json)"; tail -2 $OUT/bench.err | cut -c1-300
PROF_TIMEOUT=400 bash tools/run_profiles.sh headline 20 r5_s14/prof_headline "stats 3 4" > $OUT/prof_headline.log 2>&1; tail -2 $OUT/prof_headline.log
PROF_TIMEOUT=300 bash tools/run_profiles.sh headline 600 r5_s14/prof_config1 "stats 3 4" --net mini --games 4096 --sims 200 > $OUT/prof_config1.log 2>&1; tail -2 $OUT/prof_config1.log
bash tools/sessions/r5_s11.sh > $OUT/s11.log 2>&1; tail -12 $OUT/s11.log
find "$OUT" -name "*_kernel_trace.csv" -delete; find "$OUT" -name "*_counter_collection.csv" -delete
