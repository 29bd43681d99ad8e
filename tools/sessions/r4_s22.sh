#!/bin/bash
# Round 4, GPU session 28: counters of the solver-bound regime - k_tree_par<true> on the staggered configs[1] batch with mini.yml as shipped (two-kernel pipeline).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r4_prof_solver_bound; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export RAZ_BENCH_MINI_SHIPPED=1   # the headline leg on mini.yml's play section as shipped (4 in flight, thinking_loop 2, solver from turn 50)
BENCH="python $ROOT/bench.py --net mini --games 4096 --sims 200 --steps 40 --warmup 5 --no-cpu-baseline --no-spotcheck --no-extra-legs --full-out $OUT/bench_full.json"
SETS=("" \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F32" \
  "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES" \
  "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU")
for P in 1 2 3; do
  timeout 420 rocprofv3 --pmc ${SETS[$P]} --kernel-trace --output-format csv -d "$OUT/pmc$P" -- $BENCH < /dev/null > "$OUT/pmc$P.log" 2>&1
  echo "pmc$P rc=$?"
done
cd "$ROOT" && python tools/pmc_summary.py "$OUT" "$OUT/summary" > "$OUT/summary_pmc.txt" 2>&1; echo "summary rc=$?"
find "$OUT" -name "*_kernel_trace.csv" -delete; find "$OUT" -name "*_counter_collection.csv" -delete
python3 - <<PY
import json
d = json.load(open("$OUT/summary_pmc_per_dispatch.json"))
for k, v in d.items():
    if k.startswith("k_tree_par@") or k == "k_tree_par":
        print(k, json.dumps(v))
PY
tail -3 $OUT/pmc3.log | cut -c1-300
