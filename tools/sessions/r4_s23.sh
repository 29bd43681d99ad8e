#!/bin/bash
# Round 4, GPU session 29: A/B - non-exact (in-simulation) solves on the wave-uniform scalar search instead of the lane-parallel one.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=$PWD/gpurun_out/r4_s29; mkdir -p $O
RAZ_EXTRA_FLAGS="-DRAZ_SOLVER_NONEXACT_SCALAR=1" python reversi-alpha-zero_amd/build.py > $O/build_ab.log 2>&1; echo "build rc=$?"
timeout 600 python bench.py --steps 5 --warmup 2 --no-whole-games --no-cpu-baseline --no-spotcheck --legs config1_mini_yml_as_shipped_two_kernel_pipeline,ch5_yml_as_shipped --full-out $O/bench_full_scalar_nonexact.json > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"
python3 - <<PY
import json
d = json.load(open("$O/bench_full_scalar_nonexact.json"))
v = d.get("config1_mini_yml_as_shipped_two_kernel_pipeline") or {}
print("scalar non-exact: mini as shipped", {x: v.get(x) for x in ("value", "games_per_hour", "steps", "ms_per_step", "error")})
a = d.get("ch5_yml_as_shipped") or {}
print("scalar non-exact: ch5 as shipped", {x: a.get(x) for x in ("value", "ms_per_step", "k_tree_par_ms_per_step", "error")})
PY
python reversi-alpha-zero_amd/build.py > $O/build_default.log 2>&1; echo "rebuild rc=$?"
