#!/bin/bash
# Round 4, GPU session 17: suspended solves at the smallest budget on the four tree kernels; the updated range-flag test.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=$PWD/gpurun_out/r4_s17; mkdir -p $O
timeout 1200 python -m pytest tests/test_engine_gpu.py -q -m gpu -k "smallest_budget or range_flag or leave_the_f16" --durations=8 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -16 $O/pytest.log
