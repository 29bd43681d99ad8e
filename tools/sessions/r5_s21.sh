#!/bin/bash
# Round 5, GPU session 21: smoke() and the worker / player GPU tests on the final sources (the solver pool's round runs beside the tree
# launches under continuous batching: the worker's files must still be the oracle's rows).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r5_s21; mkdir -p $OUT
cd $ROOT
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log | cut -c1-400
timeout 80 python -m pytest tests/test_engine_gpu.py -q -m gpu -x -k "worker_files or player_facade or reference_golden_games" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest.log
