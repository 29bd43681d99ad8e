#!/bin/bash
# Round 4, GPU session 18: the N > 1 path of bench.py on the 2-rank shared-GPU gloo rig (final build), then rocprofv3 kernel
# statistics of the as-shipped leg (k_tree_par<true> with the budgeted solver).
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=$PWD/gpurun_out/r4_s18; mkdir -p $O
RAZ_BENCH_SHARED_GPU=1 timeout 600 python bench.py --gpus 2 --games 1024 --steps 5 --warmup 2 --tree-warm 8 --full-out $O/bench_2rank_rig_full.json > $O/bench_2rank_rig.json 2> $O/bench_2rank_rig.err
echo "2rank rc=$?"; tail -3 $O/bench_2rank_rig.err; wc -c $O/bench_2rank_rig.json; python3 -c "
import json; d=json.load(open('$O/bench_2rank_rig_full.json')); print(d['n_gpus'], d['value'], d.get('record_gather'), json.dumps(d.get('config4_acceptance'))[:600])"
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/as_shipped_stats -- python ${GRAFT_REPO_ROOT:-/root/repo}/bench.py --steps 5 --warmup 2 --no-whole-games --no-cpu-baseline --no-spotcheck --legs ch5_yml_as_shipped --full-out $O/bench_as_shipped_under_rocprof_full.json > $O/as_shipped_stats.log 2>&1
echo "rocprof rc=$?"; find $O/as_shipped_stats -name "*kernel_stats.csv" | head -2; head -12 $(find $O/as_shipped_stats -name "*kernel_stats.csv" | head -1) | cut -c1-200
find $O -name "*_kernel_trace.csv" -delete
