#!/bin/bash
# Round 5, GPU session 2: first hardware run of the pooled end-game solver - the solver-involving GPU tests, then the as-shipped legs.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r5_s2; mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -q -m gpu -x -k "solve or solver or eval or evaluate or series or fused" > $OUT/pytest_solver.log 2>&1
echo "pytest rc=$?"; tail -5 $OUT/pytest_solver.log
timeout 700 python bench.py --no-cpu-baseline --no-whole-games --no-spotcheck --steps 5 --warmup 2 --legs config1_mini_yml_as_shipped,config1_mini_yml_as_shipped_two_kernel_pipeline,ch5_yml_as_shipped --full-out $OUT/as_shipped_legs_full.json > $OUT/as_shipped_legs_line.json 2> $OUT/as_shipped_legs.err
echo "legs rc=$?"; tail -3 $OUT/as_shipped_legs.err
python - <<PY
import json
d = json.load(open("$OUT/as_shipped_legs_full.json"))
for k in ("config1_mini_yml_as_shipped", "config1_mini_yml_as_shipped_two_kernel_pipeline", "ch5_yml_as_shipped"):
    v = d.get(k, {})
    print(k, {x: v.get(x) for x in ("value", "games_per_hour", "steps", "ms_per_step", "k_tree_avg_ms", "error", "parity_spotcheck", "k_tree_par_ms_per_step")})
PY
