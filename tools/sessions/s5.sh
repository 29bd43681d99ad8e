#!/bin/bash
# GPU session 5 (round 3): the headline workload played to the END on the round-3 build - 8192 complete configs[2] games, one batch.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/s5; mkdir -p $O
( time timeout 1560 python tools/whole_games_config3.py --slots 8192 --games 8192 --leaf-cache-log2 26 --progress $O/progress.log > $O/whole_games_8192.json 2> $O/whole.err ) 2> $O/whole.time
echo "rc=$?"; tail -3 $O/whole.err; cat $O/whole.time; tail -2 $O/progress.log
