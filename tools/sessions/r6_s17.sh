#!/bin/bash
# Round 6, GPU session 17: the in-wave forward of the fused kernels with the first residual block's weights requested before the input
# conv is computed (-DRAZ_NET16_EARLY_WEIGHTS=1: 36 more registers live over the input conv, k_tree_net<false> 21 -> 92 spilled registers)
# against the layer requesting them where it starts - configs[1] whole games, alternating.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r6_s17; mkdir -p $OUT
cd $ROOT
RAZ_LIB_PATH=$ROOT/build/variants/libraz_e1.so timeout 600 python -m pytest tests/test_zz_fused_gpu.py -q -m gpu -x > $OUT/pytest_fused_e1.log 2>&1; echo "pytest fused (e1) rc=$?"; tail -2 $OUT/pytest_fused_e1.log
LEGS=config1_4096x200_mini,config1_mini_yml_parallel_search_num_4
for round in 1 2; do
for v in e0 e1; do
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
