#!/bin/bash
# Round 4, GPU session 31: worker-level GPU tests on the final build (three-ply solver tasks, fused kernels also with the solver on).
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=$PWD/gpurun_out/r4_s31; mkdir -p $O
timeout 240 python -m pytest tests/test_worker_scale_gpu.py tests/test_multirank_gpu.py tests/test_engine_gpu.py tests/test_engine_par_gpu.py -x -q -m gpu -k "worker or multirank or nccl or series or carried" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
