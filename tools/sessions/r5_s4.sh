#!/bin/bash
# Round 5, GPU session 4: where a solver-phase step's time goes (kernel trace of one as-shipped leg), and smaller pools / one slice.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r5_s4; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $ROOT/tools/sessions/quick_solver_ab.py "128,1024,0" > $OUT/stats.log 2>&1
echo "stats rc=$?"; tail -2 $OUT/stats.log | cut -c1-400
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
find $OUT/stats -name "*_kernel_trace.csv" -delete
head -12 $OUT/kernel_stats.csv | cut -c1-200
cd $ROOT
timeout 900 python tools/sessions/quick_solver_ab.py "128,512,0;128,256,0;256,1024,0;128,1024,0,1;128,512,0,1;256,512,0,1;128,1024,1" > $OUT/solver_ab.jsonl 2> $OUT/solver_ab.err
echo "rc=$?"; tail -2 $OUT/solver_ab.err
