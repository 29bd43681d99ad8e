#!/bin/bash
# Evidence run on the GPU box (gpurun): rocprofv3 kernel statistics and the PMC passes of
# /opt/skills/guides/MI355X_MICROARCH.md (one counter set per pass, --kernel-trace only) for the
# self-play bench or the bitboard sweep, written under gpurun_out/<name>/ and summarised by tools/pmc_summary.py
# (which REFUSES to write a traffic figure unless both the FETCH_SIZE and the WRITE_SIZE pass are there).
#   usage: tools/run_profiles.sh headline [steps] [name] [passes: "stats 1 2 3 4"] [extra bench.py args...]
#          tools/run_profiles.sh sweep    [boards] [name] [passes: "stats 1 3 4"]
# headline = BASELINE configs[2], the headline leg only (add "--net mini --games 4096 --sims 200" for the configs[1] kernels)
set -u
MODE=${1:-headline}
ARG=${2:-20}
NAME=${3:-prof_$MODE}
PASSES=${4:-"stats 1 2 3 4"}
shift; shift; shift; shift
EXTRA="$*"
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$NAME
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
if [ "$MODE" = "sweep" ]; then
  BENCH="python $ROOT/tools/bench_sweep.py --boards $ARG --steps 6 --warmup 2 --no-cpu-baseline"
  SUMMARY_ARGS="--last 6 --sweep $ARG"
else
  BENCH="python $ROOT/bench.py --steps $ARG --warmup 5 --no-cpu-baseline --no-extra-legs --no-spotcheck $EXTRA"
  SUMMARY_ARGS=""
fi
SETS=("" \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F32" \
  "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES" \
  "FETCH_SIZE" "WRITE_SIZE")
for P in $PASSES; do
  if [ "$P" = "stats" ]; then
    timeout ${PROF_TIMEOUT:-700} rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $BENCH < /dev/null > "$OUT/stats.log" 2>&1
    echo "stats rc=$?"
  else
    timeout ${PROF_TIMEOUT:-700} rocprofv3 --pmc ${SETS[$P]} --kernel-trace --output-format csv -d "$OUT/pmc$P" -- $BENCH < /dev/null > "$OUT/pmc$P.log" 2>&1
    echo "pmc$P rc=$?"
  fi
done
# gpurun copies back at most 64 MiB: summarise here and drop the per-dispatch traces
cd "$ROOT" && python tools/pmc_summary.py "$OUT" "$OUT/summary" $SUMMARY_ARGS > "$OUT/summary_pmc.txt" 2>&1
echo "summary rc=$?"
find "$OUT" -name "*_kernel_trace.csv" -delete
find "$OUT" -name "*_counter_collection.csv" -delete
find "$OUT" -name "*.csv" | head -40
