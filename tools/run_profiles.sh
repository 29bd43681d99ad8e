#!/bin/bash
# Evidence run on the GPU box (gpurun): rocprofv3 kernel statistics and the PMC passes of
# /opt/skills/guides/MI355X_MICROARCH.md (one counter set per pass, --kernel-trace only) for the
# self-play bench, written under gpurun_out/prof_final/.  Summarise afterwards with tools/pmc_summary.py.
#   usage: tools/run_profiles.sh [steps]
set -u
STEPS=${1:-600}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_final
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps $STEPS --warmup 20 --no-cpu-baseline"
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $BENCH < /dev/null > "$OUT/stats.log" 2>&1
echo "stats rc=$?"
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F32" \
           "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$OUT/pmc$i" -- $BENCH < /dev/null > "$OUT/pmc$i.log" 2>&1
  echo "pmc$i rc=$?"
done
find "$OUT" -name "*.csv" | head -40
