#!/bin/bash
# Round 6, GPU session 11: the scalar in-place solver's reach (<= 4 / 5 / 6 empties) and the lane memo's (>= 7 / 6), as-shipped legs.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r6_s11; mkdir -p $OUT
cd $ROOT
V=$ROOT/reversi-alpha-zero_amd/csrc/variants
for lib in library scalar_empties_5 scalar_empties_6 lane_memo_empties_6 library scalar_empties_5; do
  if [ $lib = library ]; then unset RAZ_LIB_PATH; else export RAZ_LIB_PATH=$V/libraz_$lib.so; fi
  timeout 200 python tools/sessions/quick_solver_ab.py "0,0,0;0,0,1;0,0,0,0,3,1" 2>> $OUT/ab.err | sed "s/^{/{\"lib\": \"$lib\", /" >> $OUT/reach_ab.jsonl
done
unset RAZ_LIB_PATH
python - <<PY
import json
for line in open("$OUT/reach_ab.jsonl"):
    try: d = json.loads(line)
    except Exception: continue
    sp = d.get("solver_pool") or {}
    print(d["lib"], {k: d.get(k) for k in ("fused", "continuous")}, "sims/s %.2f M" % (d["sims_per_s"] / 1e6), "steps", d.get("steps"), "requests", sp.get("requests_posted"), "most of one game", sp.get("most_requests_of_one_game"), "rounds/answer", sp.get("pool_rounds_per_answer"))
PY
tail -2 $OUT/ab.err | cut -c1-200
