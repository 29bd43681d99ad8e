#!/bin/bash
# Round 6, GPU session 4: (a) the bit-sliced sweep kernels (csrc/raz_sweep_sliced.h): parity tests with every superblock sliced, then the
# sweep bench A/B - sliced (default: n >= 2^21) against the board-per-lane kernels (RAZ_SWEEP_SLICED_MIN huge); (b) the hand-off with the
# pool side's agent-scope forms chosen at run time: A/B against the round-5 library again; (c) the worker leg with per-block timings.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r6_s4; mkdir -p $OUT
cd $ROOT
V=$ROOT/reversi-alpha-zero_amd/csrc/variants
RAZ_SWEEP_SLICED_MIN=2048 timeout 900 python -m pytest tests/test_sweep_gpu.py -q -m gpu -x > $OUT/pytest_sweep_sliced.log 2>&1; echo "pytest sweep (every superblock sliced) rc=$?"; tail -3 $OUT/pytest_sweep_sliced.log
timeout 600 python -m pytest tests/test_sweep_gpu.py -q -m gpu -x > $OUT/pytest_sweep_default.log 2>&1; echo "pytest sweep (default threshold) rc=$?"; tail -2 $OUT/pytest_sweep_default.log
for mode in sliced board_per_lane sliced board_per_lane; do
  if [ $mode = board_per_lane ]; then export RAZ_SWEEP_SLICED_MIN=99999999999; else unset RAZ_SWEEP_SLICED_MIN; fi
  for boards in 16777216 67108864; do
    timeout 300 python tools/bench_sweep.py --boards $boards --steps 10 --warmup 3 --no-cpu-baseline 2>> $OUT/sweep.err | sed "s/^{/{\"mode\": \"$mode\", /" >> $OUT/sweep_ab.jsonl
  done
done
unset RAZ_SWEEP_SLICED_MIN
python - <<PY
import json
for line in open("$OUT/sweep_ab.jsonl"):
    try: d = json.loads(line)
    except Exception: continue
    r, l = d["roofline"], d["k_legal_moves"]
    print(d["mode"], d["config"]["workload"][:40], "k_step %.3f ms %.0f GB/s frac %.3f | k_legal %.3f ms %.0f GB/s frac %.3f" % (r.get("avg_kernel_ms", 0), r["achieved"], r["frac"], l.get("avg_kernel_ms", 0), l["achieved"], l["frac"]))
PY
tail -2 $OUT/sweep.err | cut -c1-300
for lib in new round5 new round5; do
  if [ $lib = round5 ]; then export RAZ_LIB_PATH=$V/libraz_round5_handoff.so; else unset RAZ_LIB_PATH; fi
  timeout 300 python tools/sessions/quick_solver_ab.py "0,0,0;0,0,1;0,0,0,0,3,1" 2>> $OUT/ab.err | sed "s/^{/{\"lib\": \"$lib\", /" >> $OUT/handoff_ab.jsonl
done
unset RAZ_LIB_PATH
python - <<PY
import json
for line in open("$OUT/handoff_ab.jsonl"):
    try: d = json.loads(line)
    except Exception: continue
    print(d["lib"], {k: d.get(k) for k in ("fused", "every", "continuous")}, "sims/s %.2f M" % (d["sims_per_s"] / 1e6))
PY
timeout 400 python bench.py --no-cpu-baseline --no-whole-games --legs worker_end_to_end_config1 --full-out $OUT/bench_worker_full.json > $OUT/bench_worker.json 2> $OUT/bench_worker.err; echo "bench worker rc=$?"
python - <<PY
import json
d = json.load(open("$OUT/bench_worker_full.json"))
w = d.get("worker_end_to_end_config1", {})
print({k: w.get(k) for k in ("seconds", "games_written", "games_per_hour_including_emission", "writer_busy_share_of_the_run", "blocks", "main_thread_seconds", "engine_level_of_the_last_block", "engine_level_of_all_blocks", "error")})
for b in w.get("blocks_detail", []): print(b)
PY
timeout 600 python -m pytest tests/test_engine_gpu.py tests/test_engine_par_gpu.py tests/test_zz_fused_gpu.py tests/test_continuous_gpu.py -q -m gpu -x -k "solver or solve" > $OUT/pytest_solver.log 2>&1; echo "pytest solver rc=$?"; tail -2 $OUT/pytest_solver.log
