#!/bin/bash
# Round 4, GPU session 32: smoke() and a headline-only bench line on the final build.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=$PWD/gpurun_out/r4_s32; mkdir -p $O
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 100 python bench.py --steps 10 --warmup 3 --no-extra-legs --no-cpu-baseline --full-out $O/bench_full.json > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-200 $O/bench_line.json
