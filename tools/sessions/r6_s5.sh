#!/bin/bash
# Round 6, GPU session 5: (a) k_step_sliced compiled for two waves per SIMD (256 registers, 234 spilled to scratch) against the board-per-lane
# kernel; (b) the worker leg after the per-id stat() calls were taken out of the block's start (simulation_nums_of_ids).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r6_s5; mkdir -p $OUT
cd $ROOT
RAZ_SWEEP_SLICED_MIN=2048 timeout 300 python -m pytest tests/test_sweep_gpu.py -q -m gpu -x > $OUT/pytest_sweep_sliced.log 2>&1; echo "pytest sweep (every superblock sliced) rc=$?"; tail -2 $OUT/pytest_sweep_sliced.log
for mode in sliced board_per_lane sliced board_per_lane; do
  if [ $mode = board_per_lane ]; then export RAZ_SWEEP_SLICED_MIN=99999999999; else unset RAZ_SWEEP_SLICED_MIN; fi
  timeout 300 python tools/bench_sweep.py --boards 16777216 --steps 10 --warmup 3 --no-cpu-baseline 2>> $OUT/sweep.err | sed "s/^{/{\"mode\": \"$mode\", /" >> $OUT/sweep_ab.jsonl
done
unset RAZ_SWEEP_SLICED_MIN
python - <<PY
import json
for line in open("$OUT/sweep_ab.jsonl"):
    try: d = json.loads(line)
    except Exception: continue
    r, l = d["roofline"], d["k_legal_moves"]
    print(d["mode"], "k_step %.3f ms %.0f GB/s frac %.3f | k_legal %.3f ms %.0f GB/s frac %.3f" % (r.get("avg_kernel_ms", 0), r["achieved"], r["frac"], l.get("avg_kernel_ms", 0), l["achieved"], l["frac"]))
PY
timeout 400 python bench.py --no-cpu-baseline --no-whole-games --legs worker_end_to_end_config1 --full-out $OUT/bench_worker_full.json > $OUT/bench_worker.json 2> $OUT/bench_worker.err; echo "bench worker rc=$?"
python - <<PY
import json
d = json.load(open("$OUT/bench_worker_full.json"))
w = d.get("worker_end_to_end_config1", {})
print({k: w.get(k) for k in ("seconds", "games_written", "games_per_hour_including_emission", "writer_busy_share_of_the_run", "blocks", "main_thread_seconds", "engine_level_of_the_last_block", "engine_level_of_all_blocks", "error")})
e = (w.get("engine_level_of_all_blocks") or {}).get("games_per_hour") or 1
print("end to end / engine level of all blocks: %.3f" % (w.get("games_per_hour_including_emission", 0) / e))
for b in w.get("blocks_detail", []): print(b)
PY
tail -3 $OUT/bench_worker.err | cut -c1-300
