#!/bin/bash
# Round 5, GPU session 17: which of the "recompute instead of keeping live" measures pays on the fused kernels - A/B of libraries built with
# -DRAZ_FRESH_1 (k_tree_net: bit 0 lane id, bit 1 config words) / -DRAZ_FRESH_K (k_tree_par_net: + bit 2 lane id per round operation),
# configs[1] whole games at 1 and 4 simulations in flight, every library twice in alternating order on one box.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r5_s17; mkdir -p $OUT
cd $ROOT
LEGS=config1_4096x200_mini,config1_mini_yml_parallel_search_num_4
for round in 1 2; do
for v in head f0k0 f1k5 f2k6 f3k7 f1k4 f0k7; do
  export RAZ_LIB_PATH=$ROOT/build/variants/libraz_$v.so
  timeout 300 python bench.py --no-cpu-baseline --no-whole-games --steps 3 --warmup 1 --no-spotcheck --legs $LEGS --full-out $OUT/ab_${v}_$round.json > /dev/null 2> $OUT/ab_${v}_$round.err
  python - <<PY
import json
d = json.load(open("$OUT/ab_${v}_$round.json"))
print("$v", {k: (round(d[k]["value"] / 1e6, 2) if isinstance(d.get(k), dict) and d[k].get("value") else d.get(k)) for k in "$LEGS".split(",")})
PY
done
done
