#!/bin/bash
# Round 6, GPU session 20 (as round 5 session 11, on the final build with the gated task draws): VALU lane utilisation of k_solve_run on the leg VERDICT r4 names - mini.yml as shipped, lock-step whole games
# (two-kernel pipeline, library defaults) - one counter pass.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r6_s20; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P=$OUT/pmc_as_shipped_leg; mkdir -p $P
timeout 400 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d "$P/pmc3" -- python $ROOT/tools/sessions/quick_solver_ab.py "0,0,0" < /dev/null > "$P/pmc3.log" 2>&1
echo "pmc rc=$?"; grep sims_per_s $P/pmc3.log | cut -c1-200
cd "$ROOT" && python - <<PY
import csv, glob, json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
big = collections.defaultdict(lambda: collections.defaultdict(float))
rows = collections.defaultdict(dict)
for path in glob.glob("$P/pmc3/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        k = "k_solve_run" if "k_solve_run" in name else ("k_solve_scan" if "k_solve_scan" in name else ("k_tree_par" if "k_tree_par" in name else None))
        if not k: continue
        rows[(k, r["Dispatch_Id"])][r["Counter_Name"]] = rows[(k, r["Dispatch_Id"])].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
out = {}
for (k, d), c in rows.items():
    for name, v in c.items(): acc[k][name] += v
    n[k] += 1
    if c.get("SQ_INSTS_VALU", 0) > 1e6:   # launches that did real work
        for name, v in c.items(): big[k][name] += v
        n[k + ":busy"] += 1
for k in acc:
    a = acc[k]; b = big[k]
    out[k] = {"dispatches": n[k], "sums": dict(a), "valu_lane_utilisation": a["SQ_THREAD_CYCLES_VALU"] / (64.0 * a["SQ_ACTIVE_INST_VALU"]) if a.get("SQ_ACTIVE_INST_VALU") else None,
              "dispatches_with_over_1e6_valu_instructions": n[k + ":busy"],
              "valu_lane_utilisation_of_those": b["SQ_THREAD_CYCLES_VALU"] / (64.0 * b["SQ_ACTIVE_INST_VALU"]) if b.get("SQ_ACTIVE_INST_VALU") else None,
              "wait_any_share_of_wave_cycles": a["SQ_WAIT_ANY"] / a["SQ_WAVE_CYCLES"] if a.get("SQ_WAVE_CYCLES") else None}
json.dump(out, open("$OUT/as_shipped_leg_pmc_summary.json", "w"), indent=1)
print(json.dumps({k: {x: v[x] for x in v if x != "sums"} for k, v in out.items()}, indent=1))
PY
rm -rf $P/pmc3
