#!/bin/bash
# Round 4, GPU session 19: the as-shipped leg's played-on rate at two solver budgets.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=$PWD/gpurun_out/r4_s19; mkdir -p $O
for B in 0 1024; do
RAZ_BENCH_SOLVER_BUDGET=$B timeout 600 python bench.py --steps 8 --warmup 3 --no-whole-games --no-cpu-baseline --legs ch5_yml_as_shipped --full-out $O/bench_as_shipped_full_b$B.json > $O/bench_as_shipped_b$B.json 2> $O/bench_as_shipped_b$B.err; echo "bench budget=$B rc=$?"; python3 -c "
import json; d=json.load(open('$O/bench_as_shipped_full_b$B.json'))['ch5_yml_as_shipped']; print({k: d[k] for k in ('value','ms_per_step','k_tree_par_ms_per_step')}); p=d.get('parity_spotcheck',{}); print(p.get('result'), p.get('untimed_steps_played_on'), p.get('played_on'))"; tail -2 $O/bench_as_shipped_b$B.err
done
