#!/bin/bash
# Round 5, GPU session 3: what bounds the pooled solver on mini.yml as shipped - budget / pool size A/B with the pool's own counters.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r5_s3; mkdir -p $OUT
cd $ROOT
timeout 900 python tools/sessions/quick_solver_ab.py "0,0,0;128,0,0;64,0,0;128,1024,0;128,4096,0;128,0,1;1024,0,0" > $OUT/solver_ab.jsonl 2> $OUT/solver_ab.err
echo "rc=$?"; cat $OUT/solver_ab.jsonl | cut -c1-900; tail -3 $OUT/solver_ab.err
