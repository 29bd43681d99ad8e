#!/bin/bash
# Round 5, GPU session 15: the new pool-size GPU test, smoke() with its solver-pool games, the worker end to end with the longer poll interval.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r5_s15; mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_engine_gpu.py -q -m gpu -k "solver_pool_of_any_size" > $OUT/pytest_pool.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_pool.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log | cut -c1-500
timeout 900 python bench.py --no-cpu-baseline --no-whole-games --steps 5 --warmup 2 --no-spotcheck --legs worker_end_to_end_config1 --full-out $OUT/legs_full.json > $OUT/legs_line.json 2> $OUT/legs.err
echo "legs rc=$? lines=$(wc -l < $OUT/legs_line.json)"
python - <<PY
import json
d = json.load(open("$OUT/legs_full.json"))
v = d["worker_end_to_end_config1"]
print(json.dumps({x: v.get(x) for x in v if x not in ("workload",)})[:1500]); print("bench_wall_seconds", d.get("bench_wall_seconds"))
PY
