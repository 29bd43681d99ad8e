#!/bin/bash
# Round 6, GPU session 2: (a) the litmus again with true plain loads (inline asm; the first version's `volatile` loads were sc0 sc1);
# (b) the hardened hand-off (agent-scope words on both sides) against the round-5 library on the three as-shipped legs, same box;
# (c) solver parity tests + the trained-net test on the new library; (d) RAZ_SOLVER_PROBE_AT_DRAW=1 on the new protocol: still a stall?
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r6_s2; mkdir -p $OUT
cd $ROOT
V=$ROOT/reversi-alpha-zero_amd/csrc/variants
timeout 120 tools/litmus_slot_handoff 512 > $OUT/litmus_slot_handoff.json 2> $OUT/litmus.err; echo "litmus rc=$?"
python - <<PY
import json
d = json.load(open("$OUT/litmus_slot_handoff.json"))
for r in d["polling_reader"]:
    if r["stale_keys"] or r["tag_never_seen"]: print("POLL", r["chip"][:4], r["pairing"][:9], "|", r["publish"][:30], "|", r["read"][:24], "| stale", r["stale_keys"], "late", r["tag_never_seen"], "of", int(r["slots"]))
for r in d["single_look_after_the_publisher_finished"]:
    if r["tag_unseen"] or r["tag_seen_keys_stale"]: print("LOOK", r["pairing"][:9], "|", r["publish"][:30], "|", r["read"][:24], "| delay", r["idle_s_sleep64_before_the_look"], "unseen", r["tag_unseen"], "stale", r["tag_seen_keys_stale"], "of", r["slots"])
PY
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
    print(d["lib"], {k: d.get(k) for k in ("fused", "every", "continuous")}, "sims/s %.2f M" % (d["sims_per_s"] / 1e6), "rounds/answer", sp.get("pool_rounds_per_answer"), "lane util", sp.get("lane_utilisation"))
PY
tail -2 $OUT/ab.err | cut -c1-300
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_engine_par_gpu.py tests/test_zz_fused_gpu.py tests/test_continuous_gpu.py -q -m gpu -x -k "solver or solve" > $OUT/pytest_solver.log 2>&1; echo "pytest solver rc=$?"; tail -2 $OUT/pytest_solver.log
timeout 600 python -m pytest tests/test_net_trained_gpu.py -q -m gpu -x > $OUT/pytest_trained.log 2>&1; echo "pytest trained rc=$?"; tail -3 $OUT/pytest_trained.log
RAZ_LIB_PATH=$V/libraz_probe_at_draw.so timeout 300 python tools/sessions/debug_solver_stall.py > $OUT/probe_at_draw.log 2> $OUT/probe_at_draw.err; echo "probe at draw rc=$?"; head -c 3000 $OUT/probe_at_draw.log; tail -3 $OUT/probe_at_draw.err | cut -c1-300
