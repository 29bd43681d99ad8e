#!/bin/bash
# GPU session 2 (round 3): compact nodes - the whole GPU suite, the full bench line, PMC traffic of the headline and of configs[1].
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/s2; mkdir -p $O
python -c "import sys; sys.path.insert(0,'.'); import bench, reversi_alpha_zero_amd.engine, reversi_alpha_zero_amd.worker.self_play" || exit 9
( time timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 ) 2> $O/pytest.time
echo "pytest rc=$?"; tail -6 $O/pytest_gpu.log; cat $O/pytest.time
( time timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time
echo "bench rc=$?"; tail -3 $O/bench.err; cat $O/bench.time
timeout 600 bash tools/run_profiles.sh headline 20 s2/prof_headline "stats 3 4"
tail -3 gpurun_out/s2/prof_headline/summary_pmc.txt
timeout 600 bash tools/run_profiles.sh headline 600 s2/prof_config1 "stats 1 2 3 4" --net mini --games 4096 --sims 200
tail -3 gpurun_out/s2/prof_config1/summary_pmc.txt
