#!/bin/bash
# Round 5, GPU session 6: kernel-trace summary of the as-shipped two-kernel leg on the pooled solver (where a step's time goes).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r5_s6; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python $ROOT/tools/sessions/quick_solver_ab.py "${1:-256,512,0}" > $OUT/trace.log 2>&1
echo "trace rc=$?"; grep sims_per_s $OUT/trace.log | cut -c1-300
T=$(find $OUT/trace -name "*_kernel_trace.csv" | head -1)
python $ROOT/tools/trace_summary.py $T $OUT/trace_summary.json > $OUT/trace_summary.txt 2>&1; echo "summary rc=$?"
rm -rf $OUT/trace
