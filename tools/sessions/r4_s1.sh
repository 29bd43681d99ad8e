#!/bin/bash
# Round 4, GPU session 1 (~4 min of box time, no torch): evidence for k_conv3x3_f16x3 BEFORE touching it.
#   1. tools/probe_conv: ms per layer launch on random vs zero operands (what the power limit costs), tile-round quantisation,
#      and the per-wave timeline of one launch (start-up / K loop / barrier wait / epilogue / store drain, gaps between workgroups)
#   2. three rocprofv3 counter passes on the same binary (8192 positions, random operands), per-kernel averages
#   3. an attempt at a thread trace of one CU (needs the decoder library; best effort)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=$PWD/gpurun_out/r4_s1; mkdir -p $O
timeout 120 tools/probe_conv all 20 > $O/probe_conv.jsonl 2> $O/probe_conv.err; echo "probe rc=$?"; cat $O/probe_conv.jsonl
mv gpurun_out/probe_conv_stamps_8192.bin $O/ 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters_available.txt 2>&1
WANT_A="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA GRBM_GUI_ACTIVE GRBM_COUNT"
WANT_B="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC"
WANT_C="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_VALU"
WANT_D="SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_IFETCH SQ_WAIT_IFETCH"
i=0
for WANT in "$WANT_A" "$WANT_B" "$WANT_C" "$WANT_D"; do
  i=$((i+1)); HAVE=""
  for c in $WANT; do grep -qw "$c" $O/counters_available.txt && HAVE="$HAVE $c"; done
  echo "pass $i:$HAVE"
  timeout 120 rocprofv3 --pmc $HAVE --kernel-trace --output-format csv -d $O/pmc$i -- $GRAFT_REPO_ROOT/tools/probe_conv pmc 10 > $O/pmc$i.log 2>&1
  echo "pmc$i rc=$?"
done
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $GRAFT_REPO_ROOT/tools/probe_conv pmc 10 > $O/stats.log 2>&1; echo "stats rc=$?"
python3 - $O <<'PY'
import csv, glob, json, sys, collections
O = sys.argv[1]
out = {}
for f in sorted(glob.glob(O + "/pmc*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "?")
        if "k_conv3x3" not in k: continue
        acc[r["Counter_Name"]][r.get("Dispatch_Id")].append(float(r["Counter_Value"]))
    for c, d in acc.items():
        vals = [sum(v) for v in d.values()]          # a counter row per XCD / SE: sum them per dispatch
        out[c] = {"mean_per_dispatch": sum(vals) / len(vals), "dispatches": len(vals)}
json.dump(out, open(O + "/pmc_conv_per_dispatch.json", "w"), indent=1)
print(json.dumps(out))
PY
find $O -name "*_kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
# thread trace, best effort
timeout 90 rocprofv3 --att --kernel-include-regex k_conv3x3 -d /tmp/att -- $GRAFT_REPO_ROOT/tools/probe_conv pmc 1 > $O/att.log 2>&1; echo "att rc=$?"; tail -5 $O/att.log; du -sh /tmp/att 2>/dev/null; find /tmp/att -type f | head -20 > $O/att_files.txt
rocm-smi --showclocks --showpower > $O/rocm_smi_idle.txt 2>&1
