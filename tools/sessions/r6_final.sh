#!/bin/bash
# Round 6, final GPU session on the round's FINAL kernel sources: the counter passes the bench line's traffic figures rest on
# (profiles/r6_pmc, each file carrying the sources' sha256), rocprofv3 kernel statistics of the headline command, the whole GPU suite,
# smoke(), and the default bench run in the driver's form.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r6_final; mkdir -p $OUT $ROOT/profiles/r6_pmc
cd $ROOT
PROF_TIMEOUT=200 bash tools/run_profiles.sh headline 20 r6_final/prof_headline "stats 3 4" > $OUT/prof_headline.log 2>&1; tail -2 $OUT/prof_headline.log
PROF_TIMEOUT=150 bash tools/run_profiles.sh headline 600 r6_final/prof_config1 "stats 3 4" --net mini --games 4096 --sims 200 > $OUT/prof_config1.log 2>&1; tail -2 $OUT/prof_config1.log
PROF_TIMEOUT=150 bash tools/run_profiles.sh sweep 16777216 r6_final/prof_sweep "stats 3 4" > $OUT/prof_sweep24.log 2>&1; tail -2 $OUT/prof_sweep24.log
cp $OUT/prof_sweep/stats/*/*kernel_stats.csv $OUT/sweep_2p24_kernel_stats.csv 2>/dev/null
PROF_TIMEOUT=200 bash tools/run_profiles.sh sweep 67108864 r6_final/prof_sweep "3 4" > $OUT/prof_sweep26.log 2>&1; tail -2 $OUT/prof_sweep26.log
cp $OUT/prof_headline/summary_traffic.json $ROOT/profiles/r6_pmc/headline_traffic.json
cp $OUT/prof_headline/summary_config3_traffic.json $ROOT/profiles/r6_pmc/headline_config3_traffic.json
cp $OUT/prof_headline/summary_pmc_per_dispatch.json $ROOT/profiles/r6_pmc/headline_pmc_per_dispatch.json
cp $OUT/prof_config1/summary_traffic.json $ROOT/profiles/r6_pmc/config1_traffic.json
cp $OUT/prof_config1/summary_pmc_per_dispatch.json $ROOT/profiles/r6_pmc/config1_pmc_per_dispatch.json
cp $OUT/prof_sweep/sweep_traffic.json $ROOT/profiles/r6_pmc/sweep_traffic.json
mkdir -p $OUT/r6_pmc; cp $ROOT/profiles/r6_pmc/*.json $OUT/r6_pmc/
find $OUT -name "*kernel_stats.csv" | head
find "$OUT" -name "*_kernel_trace.csv" -delete; find "$OUT" -name "*_counter_collection.csv" -delete
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log | cut -c1-300
timeout 1500 python -m pytest tests -q -m gpu -x > $OUT/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; tail -3 $OUT/pytest_gpu.log
timeout 1200 python bench.py --full-out $OUT/bench_r6_default_run_full.json > $OUT/bench_r6_default_run_line.json 2> $OUT/bench.err; echo "bench rc=$?"; wc -c $OUT/bench_r6_default_run_line.json; cut -c1-2500 $OUT/bench_r6_default_run_line.json
