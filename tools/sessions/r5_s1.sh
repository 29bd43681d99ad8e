#!/bin/bash
# Round 5, GPU session 1: (a) socket power / clock / PPT telemetry of the headline conv kernel on random vs zero operands (VERDICT r4 next #4a);
# (b) the three counter passes of the solver-bound regime retaken on the FINAL round-4 build (three-ply tasks) - the starting point of the
# pooled solver; (c) the as-shipped legs on the same box for later A/B.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r5_s1; mkdir -p $OUT
cd $ROOT
python tools/smi_sampler.py --out $OUT/conv_power_smi_samples.jsonl --hz 20 -- tools/probe_conv power 6 8192 > $OUT/conv_power_telemetry.jsonl 2> $OUT/conv_power.err
echo "power rc=$?"; tail -6 $OUT/conv_power_telemetry.jsonl | cut -c1-600
python tools/smi_sampler.py --out $OUT/conv_power_smi_samples_7501.jsonl --hz 20 -- tools/probe_conv power 4 7501 > $OUT/conv_power_telemetry_7501.jsonl 2>> $OUT/conv_power.err
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
find "$P" -name "*_kernel_trace.csv" -delete; find "$P" -name "*_counter_collection.csv" -delete
python3 - <<PY
import json
d = json.load(open("$P/summary_pmc_per_dispatch.json"))
for k, v in d.items():
    if k.startswith("k_tree_par"):
        print(k, json.dumps(v))
PY
unset RAZ_BENCH_MINI_SHIPPED
cd $ROOT && timeout 600 python bench.py --no-cpu-baseline --no-whole-games --no-spotcheck --steps 5 --warmup 2 --legs config1_mini_yml_as_shipped,config1_mini_yml_as_shipped_two_kernel_pipeline,config1_4096x200_mini --full-out $OUT/as_shipped_legs_full.json > $OUT/as_shipped_legs_line.json 2> $OUT/as_shipped_legs.err
echo "legs rc=$?"; cut -c1-1500 $OUT/as_shipped_legs_line.json
