#!/bin/bash
# Round 5, GPU session 13: the evaluation cache's admission rule (positions of at most N discs) on the headline window: 24 (default), 32, 44.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r5_s13; mkdir -p $OUT
cd $ROOT
for V in "27 32" "28 44"; do
  set -- $V
  RAZ_BENCH_CACHE_LOG2=$1 RAZ_BENCH_CACHE_DISCS=$2 timeout 400 python bench.py --no-cpu-baseline --no-spotcheck --steps 5 --warmup 2 --window-seconds 45 --legs none --full-out $OUT/full_$2.json > $OUT/line_$2.json 2> $OUT/err_$2.txt
  python - <<PY
import json
w = json.load(open("$OUT/full_$2.json"))["whole_games_measured"]
print("$V", {k: w[k] for k in ("value", "sims_per_net_evaluation", "ms_per_step", "leaf_cache", "games_per_hour")})
PY
done
