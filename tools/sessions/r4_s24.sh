#!/bin/bash
# Round 4, GPU session 30: the lane-parallel solver with THREE-ply tasks: parity on the device, then the two as-shipped legs.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=$PWD/gpurun_out/r4_s30; mkdir -p $O
timeout 600 python -m pytest tests/test_oracle_solver.py tests/test_engine_gpu.py tests/test_engine_par_gpu.py tests/test_zz_fused_gpu.py -x -q -m gpu -k "solver or smallest_budget or evaluate_worker or shipped or fused" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout 600 python bench.py --steps 5 --warmup 2 --no-whole-games --no-cpu-baseline --no-spotcheck --legs config1_mini_yml_as_shipped,config1_mini_yml_as_shipped_two_kernel_pipeline,ch5_yml_as_shipped --full-out $O/bench_full.json > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"
python3 - <<PY
import json
d = json.load(open("$O/bench_full.json"))
for k in ("config1_mini_yml_as_shipped", "config1_mini_yml_as_shipped_two_kernel_pipeline"):
    v = d.get(k) or {}
    print(k, {x: v.get(x) for x in ("value", "games_per_hour", "steps", "ms_per_step", "error")})
a = d.get("ch5_yml_as_shipped") or {}
print("ch5 as shipped", {x: a.get(x) for x in ("value", "ms_per_step", "k_tree_par_ms_per_step", "error")}, (a.get("same_with_the_solver_off") or {}).get("value"))
PY
