#!/bin/bash
# Round 6, GPU session 7: the win/loss solves' task window (RAZ_SOLVER_NE_WINDOW: the open root moves whose tasks are handed out) - 2 (the
# library) against 0 (everything at once: rounds 4-5), 1 and 3 on the three as-shipped legs; the solver parity tests; the worker leg with
# pieces of 4096 games.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r6_s7; mkdir -p $OUT
cd $ROOT
V=$ROOT/reversi-alpha-zero_amd/csrc/variants
timeout 600 python -m pytest tests/test_engine_gpu.py tests/test_engine_par_gpu.py tests/test_zz_fused_gpu.py tests/test_continuous_gpu.py -q -m gpu -x -k "solver or solve" > $OUT/pytest_solver.log 2>&1; echo "pytest solver rc=$?"; tail -2 $OUT/pytest_solver.log
for lib in 2 0 1 3 2 0; do
  if [ $lib = 2 ]; then unset RAZ_LIB_PATH; else export RAZ_LIB_PATH=$V/libraz_ne_window_$lib.so; fi
  timeout 300 python tools/sessions/quick_solver_ab.py "0,0,0;0,0,1;0,0,0,0,3,1" 2>> $OUT/ab.err | sed "s/^{/{\"window\": $lib, /" >> $OUT/window_ab.jsonl
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
timeout 400 python bench.py --no-cpu-baseline --no-whole-games --legs worker_end_to_end_config1 --full-out $OUT/bench_worker_full.json > $OUT/bench_worker.json 2> $OUT/bench_worker.err; echo "bench worker rc=$?"
python - <<PY
import json
d = json.load(open("$OUT/bench_worker_full.json"))
w = d.get("worker_end_to_end_config1", {})
print({k: w.get(k) for k in ("seconds", "games_written", "games_per_hour_including_emission", "writer_busy_share_of_the_run", "blocks", "pieces_handed_to_the_writer", "main_thread_seconds", "engine_level_of_all_blocks", "error")})
e = (w.get("engine_level_of_all_blocks") or {}).get("games_per_hour") or 1
print("end to end / engine level of all blocks: %.3f" % (w.get("games_per_hour_including_emission", 0) / e))
for b in w.get("blocks_detail", [])[:3]: print(b)
PY
