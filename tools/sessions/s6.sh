#!/bin/bash
# GPU session 6 (round 3): the final build - smoke() and the engine parity tests that run the gamma sampler's and the tree kernels' changed code.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/s6; mkdir -p $O
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 ) 2> $O/smoke.time
echo "smoke rc=$?"; tail -2 $O/smoke.log
( time timeout 420 python -m pytest tests/test_engine_gpu.py tests/test_engine_par_gpu.py tests/test_continuous_gpu.py -x -q -k "dirichlet or golden or pruning or many_games or parallel_search_batch or continuous_batching_equals" > $O/pytest_subset.log 2>&1 ) 2> $O/pytest.time
echo "pytest rc=$?"; tail -4 $O/pytest_subset.log; cat $O/pytest.time
