#!/bin/bash
# Round 6, GPU session 6: streamed emission - the worker GPU tests (files byte-identical with and without it) and the worker leg.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r6_s6; mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_continuous_gpu.py tests/test_worker_scale_gpu.py tests/test_multirank_gpu.py -q -m gpu -x > $OUT/pytest_worker.log 2>&1; echo "pytest worker rc=$?"; tail -3 $OUT/pytest_worker.log
timeout 400 python bench.py --no-cpu-baseline --no-whole-games --legs worker_end_to_end_config1 --full-out $OUT/bench_worker_full.json > $OUT/bench_worker.json 2> $OUT/bench_worker.err; echo "bench worker rc=$?"
python - <<PY
import json
d = json.load(open("$OUT/bench_worker_full.json"))
w = d.get("worker_end_to_end_config1", {})
print({k: w.get(k) for k in ("seconds", "games_written", "games_per_hour_including_emission", "writer_busy_share_of_the_run", "blocks", "pieces_handed_to_the_writer", "main_thread_seconds", "engine_level_of_the_last_block", "engine_level_of_all_blocks", "error", "parity_check_files")})
e = (w.get("engine_level_of_all_blocks") or {}).get("games_per_hour") or 1
print("end to end / engine level of all blocks: %.3f" % (w.get("games_per_hour_including_emission", 0) / e))
for b in w.get("blocks_detail", []): print(b)
PY
tail -3 $OUT/bench_worker.err | cut -c1-300
