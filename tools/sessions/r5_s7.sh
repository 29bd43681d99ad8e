#!/bin/bash
# Round 5, GPU session 7: the whole GPU suite on the pooled-solver build, then the as-shipped legs with their parity checks.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r5_s7; mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -q -m gpu -x > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log
timeout 900 python bench.py --no-cpu-baseline --no-whole-games --no-spotcheck --steps 5 --warmup 2 --legs config1_mini_yml_as_shipped,config1_mini_yml_as_shipped_two_kernel_pipeline,ch5_yml_as_shipped --full-out $OUT/as_shipped_legs_full.json > $OUT/as_shipped_legs_line.json 2> $OUT/as_shipped_legs.err
echo "legs rc=$?"; tail -3 $OUT/as_shipped_legs.err
python - <<PY
import json
d = json.load(open("$OUT/as_shipped_legs_full.json"))
for k in ("config1_mini_yml_as_shipped", "config1_mini_yml_as_shipped_two_kernel_pipeline", "ch5_yml_as_shipped"):
    v = d.get(k, {})
    print(k, {x: v.get(x) for x in ("value", "games_per_hour", "steps", "ms_per_step", "k_tree_avg_ms", "error", "parity_spotcheck", "k_tree_par_ms_per_step")})
PY
