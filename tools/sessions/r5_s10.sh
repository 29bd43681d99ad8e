#!/bin/bash
# Round 5, GPU session 10: the default bench run of the final build, then the profiles the line's figures rest on: kernel statistics and
# the FETCH_SIZE / WRITE_SIZE passes of the headline command and of the configs[1] command (profiles/r5_pmc, with the kernel sources'
# sha256), and three counter passes of the solver-bound regime (k_solve_run's lane utilisation).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r5_s10; mkdir -p $OUT
cd $ROOT
timeout 900 python bench.py --full-out $OUT/bench_full.json > $OUT/bench_line.json 2> $OUT/bench.err
echo "bench rc=$? bytes=$(wc -c < $OUT/bench_line.json)"; tail -3 $OUT/bench.err | cut -c1-300
PROF_TIMEOUT=400 bash tools/run_profiles.sh headline 20 r5_s10/prof_headline "stats 3 4" > $OUT/prof_headline.log 2>&1; tail -4 $OUT/prof_headline.log
PROF_TIMEOUT=300 bash tools/run_profiles.sh headline 600 r5_s10/prof_config1 "stats 3 4" --net mini --games 4096 --sims 200 > $OUT/prof_config1.log 2>&1; tail -4 $OUT/prof_config1.log
cd /tmp && export TMPDIR=/tmp
export RAZ_BENCH_MINI_SHIPPED=1
P=$OUT/pmc_solver_bound; mkdir -p $P
BENCH="python $ROOT/bench.py --net mini --games 4096 --sims 200 --steps 40 --warmup 5 --no-cpu-baseline --no-spotcheck --no-extra-legs --full-out $P/bench_full.json"
SETS=("" \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F32" \
  "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES" \
  "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU")
for K in 1 2 3; do
  timeout 300 rocprofv3 --pmc ${SETS[$K]} --kernel-trace --output-format csv -d "$P/pmc$K" -- $BENCH < /dev/null > "$P/pmc$K.log" 2>&1
  echo "pmc$K rc=$?"
done
cd "$ROOT" && python tools/pmc_summary.py "$P" "$P/summary" > "$P/summary_pmc.txt" 2>&1; echo "summary rc=$?"
find "$OUT" -name "*_kernel_trace.csv" -delete; find "$OUT" -name "*_counter_collection.csv" -delete
python3 - <<PY
import json
d = json.load(open("$P/summary_pmc_per_dispatch.json"))
for k, v in d.items():
    if "@" not in k and (k.startswith("k_solve") or k.startswith("k_tree_par")):
        print(k, json.dumps(v))
PY
