#!/bin/bash
# Round 4, GPU session 24: the whole GPU suite, then the default bench run (the driver's form), final build.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=$PWD/gpurun_out/r4_s24; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
timeout 1200 python bench.py --full-out $O/bench_full.json > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"; wc -c $O/bench_line.json; cut -c1-600 $O/bench_line.json; tail -3 $O/bench.err
