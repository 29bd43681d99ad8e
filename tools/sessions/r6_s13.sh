#!/bin/bash
# Round 6, GPU session 13: the pool's worker waves with LDS frames sized for the configuration (6 levels = 12 KB at use_solver_turn 50) against
# the fixed 14 levels (28 KB): solver parity first, then lock-step at 1 / 2 / 3 tree launches per round and continuous batching at 2 / 3 / 4.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r6_s13; mkdir -p $OUT
cd $ROOT
V=$ROOT/reversi-alpha-zero_amd/csrc/variants
timeout 120 python tools/sessions/quick_solver_ab.py "0,0,0" > $OUT/first.jsonl 2> $OUT/first.err; rc=$?; echo "guarded first run rc=$rc"; cut -c1-200 $OUT/first.jsonl
if [ $rc -ne 0 ]; then tail -3 $OUT/first.err | cut -c1-300; exit 1; fi
timeout 400 python -m pytest tests/test_engine_gpu.py tests/test_engine_par_gpu.py tests/test_zz_fused_gpu.py tests/test_continuous_gpu.py -q -m gpu -x -k "solver or solve" > $OUT/pytest_solver.log 2>&1; echo "pytest solver rc=$?"; tail -2 $OUT/pytest_solver.log
for lib in sized fixed14 sized fixed14; do
  if [ $lib = sized ]; then unset RAZ_LIB_PATH; else export RAZ_LIB_PATH=$V/libraz_frames_14_levels.so; fi
  timeout 400 python tools/sessions/quick_solver_ab.py "0,0,0,0,1;0,0,0,0,2;0,0,0,0,3;0,0,1;0,0,0,0,2,1;0,0,0,0,3,1;0,0,0,0,4,1;0,1280,0,0,3,1" 2>> $OUT/ab.err | sed "s/^{/{\"frames\": \"$lib\", /" >> $OUT/frames_ab.jsonl
done
unset RAZ_LIB_PATH
python - <<PY
import json
for line in open("$OUT/frames_ab.jsonl"):
    try: d = json.loads(line)
    except Exception: continue
    print(d["frames"], {k: d.get(k) for k in ("waves", "fused", "every", "continuous")}, "sims/s %.2f M" % (d["sims_per_s"] / 1e6), "steps", d.get("steps"), "ms/step %.3f" % d.get("ms_per_step", 0))
PY
tail -2 $OUT/ab.err | cut -c1-200
