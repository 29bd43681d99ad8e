#!/bin/bash
# Round 4, GPU session 15: the v2 range repair tests again.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=$PWD/gpurun_out/r4_s15; mkdir -p $O
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_worker_scale_gpu.py tests/test_leaf_cache_gpu.py -q -m gpu -k "f16x3 or overflows or leaf_cache" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest.log
