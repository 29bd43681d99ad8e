#!/bin/bash
# GPU session 4 (round 3): counter traffic of the tree kernel on the headline command (full size, else 1024 games), then the
# final default bench line with every traffic file in place.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/s4; mkdir -p $O
python -c "import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tools'); import bench, reversi_alpha_zero_amd.engine" || exit 9
PROF_TIMEOUT=300 timeout 700 bash tools/run_profiles.sh headline 20 s4/prof_headline_full "3 4"
if [ ! -f $O/prof_headline_full/summary_traffic.json ]; then
  PROF_TIMEOUT=240 timeout 600 bash tools/run_profiles.sh headline 20 s4/prof_headline_1024 "3 4" --games 1024 --no-leaf-cache
fi
python - <<'PY'
import json, os
for name, games in (("prof_headline_full", 8192), ("prof_headline_1024", 1024)):
    p = f"gpurun_out/s4/{name}/summary_traffic.json"
    if os.path.exists(p):
        t = json.load(open(p))
        t["games_per_launch"] = games
        t["command"] = "tools/run_profiles.sh headline 20 ... " + ("" if games == 8192 else "--games 1024 --no-leaf-cache")
        for dst in ("profiles/r3_pmc/headline_ktree_traffic.json", "gpurun_out/s4/headline_ktree_traffic.json"):
            json.dump(t, open(dst, "w"), indent=1, sort_keys=True)
        print("k_tree traffic from", name, {k: round(v) for k, v in t["kernels"]["k_tree"].items()})
        break
else:
    print("no tree-kernel traffic profile")
PY
( time timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time
echo "bench rc=$?"; tail -3 $O/bench.err; cat $O/bench.time
