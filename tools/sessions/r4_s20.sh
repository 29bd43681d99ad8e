#!/bin/bash
# Round 4, GPU session 25: worker-level GPU tests and smoke() after the fused-kernel rule became a function.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=$PWD/gpurun_out/r4_s25; mkdir -p $O
timeout 1200 python -m pytest tests/test_worker_scale_gpu.py tests/test_multirank_gpu.py tests/test_zz_fused_gpu.py tests/test_engine_gpu.py -q -m gpu -k "worker or fused or multirank or nccl or self_worker" --durations=5 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
