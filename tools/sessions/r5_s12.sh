#!/bin/bash
# Round 5, GPU session 12: compile-time knobs of the worker loop, rebuilt on the box: slow phase every 4 / 8 / 16 iterations, lane memo from 6 / 7 empties.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r5_s12; mkdir -p $OUT
cd $ROOT
: > $OUT/ab.jsonl
for V in "-DRAZ_SOLVER_SLOW_EVERY=16" "-DRAZ_SOLVER_SLOW_EVERY=4" "-DRAZ_SOLVER_LANE_MEMO_EMPTIES=7" "-DRAZ_SOLVER_LANE_MEMO_EMPTIES=7 -DRAZ_SOLVER_SLOW_EVERY=16"; do
  RAZ_EXTRA_FLAGS="$V" python reversi-alpha-zero_amd/build.py > $OUT/build.log 2>&1 || { echo "build failed $V"; tail -3 $OUT/build.log; continue; }
  echo "{\"variant\": \"$V\"}" >> $OUT/ab.jsonl
  RAZ_BENCH_MAX_STEPS=60000 timeout 300 python tools/sessions/quick_solver_ab.py "0,0,0" >> $OUT/ab.jsonl 2>> $OUT/ab.err
done
python reversi-alpha-zero_amd/build.py > /dev/null 2>&1
grep -o '"variant": "[^"]*"\|"sims_per_s": [0-9.]*\|"steps": [0-9]*\|"pool_rounds_per_answer": [0-9.]*\|"ticks_per_wave_iteration": [0-9.]*' $OUT/ab.jsonl
