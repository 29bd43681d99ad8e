#!/bin/bash
# GPU session 1 (round 3): new worker tests + cache/continuous tests, the full default bench line, headline PMC traffic passes.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/s1; mkdir -p $O
(nproc; free -g | head -2; rocm-smi --showmeminfo vram 2>/dev/null | head -8) > $O/box.txt 2>&1
timeout 900 python -m pytest tests/test_worker_scale_gpu.py tests/test_leaf_cache_gpu.py tests/test_continuous_gpu.py -x -q > $O/pytest_subset.log 2>&1
echo "pytest rc=$?" | tee -a $O/pytest_subset.log
tail -5 $O/pytest_subset.log
( time timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time
echo "bench rc=$?"; tail -3 $O/bench.err; cat $O/bench.time
RAZ_BENCH_SHARED_GPU=1 timeout 600 python bench.py --gpus 2 --games 1024 --steps 5 --warmup 2 --tree-warm 8 > $O/bench_2rank_rig.json 2> $O/bench_2rank_rig.err
echo "2rank rc=$?"; tail -3 $O/bench_2rank_rig.err
timeout 900 bash tools/run_profiles.sh headline 20 s1/prof_headline "stats 3 4"
cat gpurun_out/s1/prof_headline/summary_pmc.txt | tail -20
