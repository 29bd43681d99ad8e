#!/bin/bash
# Round 4, GPU session 33: the large-task-tree solver test on the device.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=$PWD/gpurun_out/r4_s33; mkdir -p $O
timeout 100 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "11_to_13" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log
