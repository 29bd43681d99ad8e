#!/bin/bash
# Round 6, GPU session 15: the fused kernels with the engine's descriptor read from the kernel-argument segment where it is used
# (fresh_descriptor: RAZ_FRESH_1 / RAZ_FRESH_K bit 3) instead of held in scalar registers and parked in vector lanes for the whole launch -
# parity (the fused GPU tests on the variant library), then A/B of libraries on configs[1] whole games, alternating on one box.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r6_s15; mkdir -p $OUT
cd $ROOT
RAZ_LIB_PATH=$ROOT/build/variants/libraz_f8k15.so timeout 600 python -m pytest tests/test_zz_fused_gpu.py tests/test_engine_gpu.py -q -m gpu -x > $OUT/pytest_fused_f8k15.log 2>&1; echo "pytest fused (f8k15) rc=$?"; tail -2 $OUT/pytest_fused_f8k15.log
LEGS=config1_4096x200_mini,config1_mini_yml_parallel_search_num_4,config1_mini_yml_as_shipped
for round in 1 2; do
for v in head f8k15 f8k7 f0k15; do
  export RAZ_LIB_PATH=$ROOT/build/variants/libraz_$v.so
  timeout 400 python bench.py --no-cpu-baseline --no-whole-games --steps 3 --warmup 1 --no-spotcheck --legs $LEGS --full-out $OUT/ab_${v}_$round.json > /dev/null 2> $OUT/ab_${v}_$round.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/ab_${v}_$round.json"))
    print("$v", {k: (round(d[k]["value"] / 1e6, 2) if isinstance(d.get(k), dict) and d[k].get("value") else d.get(k)) for k in "$LEGS".split(",")})
except Exception as e:
    print("$v", "no result", e)
PY
done
done
