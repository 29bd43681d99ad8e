#!/bin/bash
# Round 4, GPU session 5: the changed GPU tests (fused kernel as the worker's default, the nccl group of one rank), then the rewritten
# bench at reduced size (every leg once), then the nccl rig of the bench.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=$PWD/gpurun_out/r4_s5; mkdir -p $O
timeout 900 python -m pytest tests/test_zz_fused_gpu.py tests/test_multirank_gpu.py tests/test_worker_scale_gpu.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
timeout 700 python bench.py --games 1024 --steps 4 --warmup 2 --window-seconds 20 --full-out $O/bench_small_full.json > $O/bench_small.json 2> $O/bench_small.err; echo "bench rc=$?"; tail -c 3000 $O/bench_small.json; tail -5 $O/bench_small.err
RAZ_BENCH_NCCL_WORLD1=1 timeout 300 python bench.py --games 512 --steps 3 --warmup 1 --no-extra-legs --no-cpu-baseline --full-out $O/bench_nccl1_full.json > $O/bench_nccl1.json 2> $O/bench_nccl1.err; echo "nccl rig rc=$?"; tail -c 1500 $O/bench_nccl1.json; tail -5 $O/bench_nccl1.err
