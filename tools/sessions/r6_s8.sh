#!/bin/bash
# Round 6, GPU session 8: the win/loss task window again after the ticket fix (past-the-end tickets go back too): one guarded run first (a
# stall costs minutes), then the solver parity tests, then the A/B 2 (library) / 0 / 1 / 3.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r6_s8; mkdir -p $OUT
cd $ROOT
V=$ROOT/reversi-alpha-zero_amd/csrc/variants
timeout 120 python tools/sessions/quick_solver_ab.py "0,0,0" > $OUT/first.jsonl 2> $OUT/first.err; rc=$?; echo "guarded first run rc=$rc"; cut -c1-400 $OUT/first.jsonl
if [ $rc -ne 0 ]; then tail -3 $OUT/first.err | cut -c1-300; exit 1; fi
timeout 400 python -m pytest tests/test_engine_gpu.py tests/test_engine_par_gpu.py tests/test_zz_fused_gpu.py tests/test_continuous_gpu.py -q -m gpu -x -k "solver or solve" > $OUT/pytest_solver.log 2>&1; echo "pytest solver rc=$?"; tail -2 $OUT/pytest_solver.log
for lib in 2 0 1 3 2 0; do
  if [ $lib = 2 ]; then unset RAZ_LIB_PATH; else export RAZ_LIB_PATH=$V/libraz_ne_window_$lib.so; fi
  timeout 200 python tools/sessions/quick_solver_ab.py "0,0,0;0,0,1;0,0,0,0,3,1" 2>> $OUT/ab.err | sed "s/^{/{\"window\": $lib, /" >> $OUT/window_ab.jsonl
done
unset RAZ_LIB_PATH
python - <<PY
import json
for line in open("$OUT/window_ab.jsonl"):
    try: d = json.loads(line)
    except Exception: continue
    sp = d.get("solver_pool") or {}
    print("window", d["window"], {k: d.get(k) for k in ("fused", "every", "continuous")}, "sims/s %.2f M" % (d["sims_per_s"] / 1e6), "steps", d.get("steps"), "rounds/answer", sp.get("pool_rounds_per_answer"), "busy lane-iterations", sp.get("busy_lane_iterations"), "util", sp.get("lane_utilisation"))
PY
tail -2 $OUT/ab.err | cut -c1-300
