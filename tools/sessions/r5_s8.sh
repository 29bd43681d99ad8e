#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r5_s8; mkdir -p $OUT
cd $ROOT
timeout 300 python tools/sessions/debug_solver_stall.py > $OUT/clean.log 2>&1; echo "clean rc=$?"; tail -3 $OUT/clean.log | cut -c1-400
timeout 400 python tools/sessions/debug_solver_stall.py --dirty > $OUT/dirty.log 2>&1; echo "dirty rc=$?"; tail -25 $OUT/dirty.log | cut -c1-900
