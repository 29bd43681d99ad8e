#!/bin/bash
# Round 4, GPU session 7: the lane-parallel solver after the memo / frame changes: parity, its own timing, the as-shipped leg.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=$PWD/gpurun_out/r4_s10; mkdir -p $O
true
true
timeout 600 python bench.py --steps 8 --warmup 3 --no-whole-games --no-cpu-baseline --no-spotcheck --legs ch5_yml_as_shipped --full-out $O/bench_as_shipped_full.json > $O/bench_as_shipped.json 2> $O/bench_as_shipped.err; echo "bench rc=$?"; python3 -c "
import json; d=json.load(open('$O/bench_as_shipped_full.json')); print(json.dumps(d.get('ch5_yml_as_shipped'))[:1600])"; tail -3 $O/bench_as_shipped.err
