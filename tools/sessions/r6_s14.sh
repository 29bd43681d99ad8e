#!/bin/bash
# Round 6, GPU session 14: ReversiEnv.step in the hybrid form (k_step_hybrid: the move per board, the legal moves after it bit-sliced,
# 256 registers + 10 KB LDS, two waves per SIMD, persistent waves that issue their next superblock's loads before the stores of the one
# at hand) - parity (one superblock per wave, and 100 waves walking all of them), then A/B against the board-per-lane kernel and k_step_sliced.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r6_s14; mkdir -p $OUT
cd $ROOT
RAZ_SWEEP_SLICED_STEP=2 RAZ_SWEEP_SLICED_MIN=2048 timeout 300 python -m pytest tests/test_sweep_gpu.py -q -m gpu -x > $OUT/pytest_sweep_hybrid.log 2>&1; echo "pytest sweep (hybrid step, every superblock) rc=$?"; tail -2 $OUT/pytest_sweep_hybrid.log
RAZ_SWEEP_HYBRID_WAVES=100 RAZ_SWEEP_SLICED_STEP=2 RAZ_SWEEP_SLICED_MIN=2048 timeout 300 python -m pytest tests/test_sweep_gpu.py -q -m gpu -x > $OUT/pytest_sweep_hybrid_100_waves.log 2>&1; echo "pytest sweep (hybrid step, 100 waves) rc=$?"; tail -2 $OUT/pytest_sweep_hybrid_100_waves.log
for boards in 16777216 67108864; do
for mode in 0 2 2s2 2s4 2s6 0 2 2s2 2s4 2s6; do
  case $mode in 2s*) export RAZ_SWEEP_HYBRID_STAGGER=${mode#2s}; m=2;; *) unset RAZ_SWEEP_HYBRID_STAGGER; m=$mode;; esac
  RAZ_SWEEP_SLICED_STEP=$m timeout 300 python tools/bench_sweep.py --boards $boards --steps 10 --warmup 3 --no-cpu-baseline 2>> $OUT/sweep.err | sed "s/^{/{\"step_form\": \"$mode\", \"boards\": $boards, /" >> $OUT/sweep_ab.jsonl
done
done
unset RAZ_SWEEP_HYBRID_STAGGER
python - <<PY
import json
for line in open("$OUT/sweep_ab.jsonl"):
    try: d = json.loads(line)
    except Exception: continue
    r = d["roofline"]
    print(d["boards"], "step_form", d["step_form"], "k_step %.3f ms %.0f GB/s frac %.3f" % (r.get("avg_kernel_ms", 0), r["achieved"], r["frac"]))
PY
tail -3 $OUT/sweep.err | cut -c1-300
