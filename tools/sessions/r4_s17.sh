#!/bin/bash
# Round 4, GPU session 21: mini.yml as shipped on the configs[1] batch, fused kernels against the two-kernel pipeline.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=$PWD/gpurun_out/r4_s22; mkdir -p $O
timeout 900 python bench.py --steps 5 --warmup 2 --no-whole-games --no-cpu-baseline --no-spotcheck --legs config1_mini_yml_as_shipped,config1_mini_yml_as_shipped_two_kernel_pipeline --full-out $O/bench_full.json > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"
python3 - <<PY
import json
d = json.load(open("$O/bench_full.json"))
for k in ("config1_mini_yml_as_shipped", "config1_mini_yml_as_shipped_two_kernel_pipeline"):
    v = d.get(k) or {}
    print(k, {x: v.get(x) for x in ("value", "games_per_hour", "steps", "ms_per_step", "error")})
PY
tail -3 $O/bench.err
