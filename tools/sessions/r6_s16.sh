#!/bin/bash
# Round 6, GPU session 16: the engine's descriptor read from the kernel-argument segment where it is used, in ALL four tree kernels
# (fresh_descriptor: k_tree / k_tree_par RAZ_FRESH_DESC, k_tree_net / k_tree_par_net bit 3 of RAZ_FRESH_1 / RAZ_FRESH_K) -
# parity (the engine GPU tests on the new library), then A/B against the old forms: headline (20 steps) + the configs[1] / mini.yml legs.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r6_s16; mkdir -p $OUT
cd $ROOT
RAZ_LIB_PATH=$ROOT/build/variants/libraz_new.so timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_engine_par_gpu.py tests/test_zz_fused_gpu.py tests/test_continuous_gpu.py -q -m gpu -x > $OUT/pytest_engine_new.log 2>&1; echo "pytest engine (new) rc=$?"; tail -2 $OUT/pytest_engine_new.log
LEGS=config1_4096x200_mini,config1_mini_yml_parallel_search_num_4,config1_mini_yml_as_shipped,config1_mini_yml_as_shipped_two_kernel_pipeline,config1_two_kernel_pipeline,config1_two_kernel_pipeline_parallel_search_num_4,config1_mini_yml_as_shipped_continuous_batching,ch5_yml_as_shipped
for round in 1 2; do
for v in old new fusedonly; do
  export RAZ_LIB_PATH=$ROOT/build/variants/libraz_$v.so
  timeout 600 python bench.py --no-cpu-baseline --no-whole-games --steps 20 --warmup 5 --no-spotcheck --legs $LEGS --full-out $OUT/ab_${v}_$round.json > /dev/null 2> $OUT/ab_${v}_$round.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/ab_${v}_$round.json"))
    leg = lambda k: (round(d[k]["value"] / 1e6, 2) if isinstance(d.get(k), dict) and d[k].get("value") else (str(d.get(k))[:60] if d.get(k) is not None else None))
    print("$v", "headline %.1f k" % (d["value"] / 1e3), {k.replace("config1_", "").replace("mini_yml_", ""): leg(k) for k in "$LEGS".split(",")})
except Exception as e:
    print("$v", "no result", e)
PY
done
done
