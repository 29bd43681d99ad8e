#!/bin/bash
# Round 4, GPU session 23: mini.yml as shipped, solver kernels built for 4 waves per SIMD (128 VGPRs, spills) against the default 2.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=$PWD/gpurun_out/r4_s23; mkdir -p $O
RAZ_EXTRA_FLAGS="-DRAZ_SOLVER_WAVES=4" python reversi-alpha-zero_amd/build.py > $O/build4.log 2>&1; echo "build4 rc=$?"
timeout 900 python bench.py --steps 5 --warmup 2 --no-whole-games --no-cpu-baseline --no-spotcheck --legs config1_mini_yml_as_shipped,config1_mini_yml_as_shipped_two_kernel_pipeline --full-out $O/bench_full_waves4.json > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"
python3 - <<PY
import json
d = json.load(open("$O/bench_full_waves4.json"))
for k in ("config1_mini_yml_as_shipped", "config1_mini_yml_as_shipped_two_kernel_pipeline"):
    v = d.get(k) or {}
    print("4 waves/SIMD", k, {x: v.get(x) for x in ("value", "games_per_hour", "steps", "ms_per_step", "error")})
PY
python reversi-alpha-zero_amd/build.py > $O/build2.log 2>&1; echo "build2 rc=$?"
