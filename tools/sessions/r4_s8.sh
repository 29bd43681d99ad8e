#!/bin/bash
# Round 4, GPU session 11: in-simulation solves on the per-launch solver budget (suspended descents): parity, then the as-shipped
# leg at three budgets.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=$PWD/gpurun_out/r4_s11; mkdir -p $O
timeout 900 python -m pytest tests/test_oracle_solver.py tests/test_engine_gpu.py -x -q -m gpu -k "device_solver or with_solver or evaluate_worker or slot" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
for B in 0 1024 4096; do
RAZ_BENCH_SOLVER_BUDGET=$B timeout 600 python bench.py --steps 8 --warmup 3 --no-whole-games --no-cpu-baseline --no-spotcheck --legs ch5_yml_as_shipped --full-out $O/bench_as_shipped_full_b$B.json > $O/bench_as_shipped_b$B.json 2> $O/bench_as_shipped_b$B.err; echo "bench budget=$B rc=$?"; python3 -c "
import json; d=json.load(open('$O/bench_as_shipped_full_b$B.json'))['ch5_yml_as_shipped']; print({k: d[k] for k in ('value','ms_per_step','k_tree_par_ms_per_step','net_forward_ms_per_step','sims_per_step')}, d['same_with_the_solver_at_the_root_only']['value'], d['same_with_the_solver_off']['value'])"; tail -2 $O/bench_as_shipped_b$B.err
done
