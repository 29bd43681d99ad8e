#!/bin/bash
# Round 6, GPU session 3: (a) RAZ_SOLVER_PROBE_AT_DRAW=1 on the ROUND-5 sources: does the stall reproduce on this box (twice, and on a
# dirty workspace)?  (b) hand-off A/B again with the drain only in waves that store keys; (c) worker end to end with worker.start()'s
# new default block (16 games per slot for a 16-filter net); (d) a longer training run (6 generations) for the net accuracy record;
# (e) the whole GPU suite on the new library.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r6_s3; mkdir -p $OUT
cd $ROOT
V=$ROOT/reversi-alpha-zero_amd/csrc/variants
for i in 1 2; do
  RAZ_LIB_PATH=$V/libraz_round5_probe_at_draw.so timeout 200 python tools/sessions/debug_solver_stall.py > $OUT/round5_probe_at_draw_$i.log 2> $OUT/round5_probe_at_draw_$i.err; echo "round-5 sources + probe at draw, run $i rc=$?"; head -c 1500 $OUT/round5_probe_at_draw_$i.log; echo
done
RAZ_LIB_PATH=$V/libraz_round5_probe_at_draw.so timeout 200 python tools/sessions/debug_solver_stall.py --dirty > $OUT/round5_probe_at_draw_dirty.log 2> $OUT/round5_probe_at_draw_dirty.err; echo "round-5 sources + probe at draw, dirty rc=$?"; head -c 1500 $OUT/round5_probe_at_draw_dirty.log; echo
RAZ_LIB_PATH=$V/libraz_probe_at_draw.so timeout 200 python tools/sessions/debug_solver_stall.py --dirty > $OUT/new_probe_at_draw_dirty.log 2> $OUT/new_probe_at_draw_dirty.err; echo "new sources + probe at draw, dirty rc=$?"; head -c 600 $OUT/new_probe_at_draw_dirty.log; echo
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
    sp = d.get("solver_pool") or {}
    print(d["lib"], {k: d.get(k) for k in ("fused", "every", "continuous")}, "sims/s %.2f M" % (d["sims_per_s"] / 1e6), "rounds/answer", sp.get("pool_rounds_per_answer"))
PY
timeout 400 python bench.py --no-cpu-baseline --no-whole-games --legs worker_end_to_end_config1,config1_continuous_batching --full-out $OUT/bench_worker_full.json > $OUT/bench_worker.json 2> $OUT/bench_worker.err; echo "bench worker rc=$?"
python - <<PY
import json
d = json.load(open("$OUT/bench_worker_full.json"))
w = d.get("worker_end_to_end_config1", {})
print({k: w.get(k) for k in ("seconds", "games_written", "games_per_hour_including_emission", "writer_busy_share_of_the_run", "blocks", "main_thread_seconds", "engine_level_of_the_last_block", "error")})
print("ratio", w.get("games_per_hour_including_emission", 0) / max(1, (w.get("engine_level_of_the_last_block") or {}).get("games_per_hour") or 1))
print({k: d.get("config1_continuous_batching", {}).get(k) for k in ("value", "games_per_hour")})
PY
timeout 1500 python tools/trained_net.py --generations 6 --games 4096 --sims 64 --steps 2500 --out $OUT/net_v2_on_a_gpu_trained_256x10_net_6_generations.json > $OUT/trained6.log 2> $OUT/trained6.err; echo "trained6 rc=$?"; tail -1 $OUT/trained6.log | cut -c1-700; grep "generation\|held-out" $OUT/trained6.err | cut -c1-300
timeout 1500 python -m pytest tests -q -m gpu -x > $OUT/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; tail -3 $OUT/pytest_gpu.log
