#!/bin/bash
# GPU session 3 (round 3): the whole GPU suite on the final kernels, the net accuracy sweep, PMC passes of the sweep kernels
# (2^24 and 2^26 boards) and of the headline.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/s3; mkdir -p $O
python -c "import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tools'); import bench, check_net_accuracy, reversi_alpha_zero_amd.engine, reversi_alpha_zero_amd.worker.self_play" || exit 9
( time timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1 ) 2> $O/pytest.time
echo "pytest rc=$?"; tail -6 $O/pytest_gpu.log; cat $O/pytest.time
( time timeout 900 python tools/check_net_accuracy.py --n 131072 --out $O/net_accuracy.json > $O/net_accuracy.out 2> $O/net_accuracy.err ) 2> $O/acc.time
echo "accuracy rc=$?"; tail -6 $O/net_accuracy.err; cat $O/acc.time
timeout 900 bash tools/run_profiles.sh sweep 16777216 s3/prof_sweep24 "stats 1 2 3 4"
tail -3 $O/prof_sweep24/summary_pmc.txt
timeout 900 bash tools/run_profiles.sh sweep 67108864 s3/prof_sweep26 "3 4"
tail -3 $O/prof_sweep26/summary_pmc.txt
timeout 1500 bash tools/run_profiles.sh headline 20 s3/prof_headline "stats 3 4"
tail -3 $O/prof_headline/summary_pmc.txt
