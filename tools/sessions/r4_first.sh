#!/bin/bash
# First GPU session of the next round (~4 min): what round 3 built after its GPU minutes were spent.
#   1. the isolated GPU test of the fused kernels and of the split-net pytest form
#   2. configs[1] whole games: classic pipeline vs k_tree_net (parallel_search_num 1) and vs k_tree_par_net (mini.yml's 4)
#   3. the headline conv kernel against its four hand-scheduled variants (bit equality + ms per forward)
#   4. counter traffic of k_tree_net on the configs[1] command (FETCH_SIZE / WRITE_SIZE passes) -> gpurun_out/r4_first/prof_fused
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r4_first; mkdir -p $O
timeout 400 python -m pytest tests/test_zz_fused_gpu.py -q -rxXs > $O/pytest_fused.log 2>&1; tail -4 $O/pytest_fused.log
for v in fused fused_par4; do timeout 200 python bench.py --config1-variant $v 2> $O/config1_$v.err | tail -1 > $O/config1_$v.json; head -c 600 $O/config1_$v.json; echo; done
timeout 200 python tools/sessions/quick_f16x3_pipe.py 2> $O/pipe.err | tail -1 | tee $O/f16x3_pipe_ab.json
PROF_TIMEOUT=200 timeout 500 bash tools/run_profiles.sh headline 200 r4_first/prof_fused "stats 3 4" --net mini --games 4096 --sims 200 --fused
cat $O/prof_fused/summary_pmc.txt 2>/dev/null | head -30
