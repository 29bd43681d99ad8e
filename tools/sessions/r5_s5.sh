#!/bin/bash
# Round 5, GPU session 5: the pool with LDS frames, one-round memo probes and 5-6-empties solves in the pool - A/B of pool size / budget.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r5_s5; mkdir -p $OUT
cd $ROOT
RAZ_BENCH_MAX_STEPS=${RAZ_BENCH_MAX_STEPS:-60000} timeout 900 python tools/sessions/quick_solver_ab.py "${1:-128,256,0;128,512,0;128,1024,0;128,2048,0;256,512,0;64,256,0;128,128,0;128,512,1;256,1024,1}" > $OUT/solver_ab.jsonl 2> $OUT/solver_ab.err
echo "rc=$?"; tail -2 $OUT/solver_ab.err
