#!/bin/bash
# Round 4, GPU session 16: counter traffic of k_tree after the 4-slot aligned table probes (headline command and the configs[1] two-kernel pipeline).
cd ${GRAFT_REPO_ROOT:-/root/repo}
PROF_TIMEOUT=400 bash tools/run_profiles.sh headline 20 r4_prof_ktree_headline "3 4"
PROF_TIMEOUT=300 bash tools/run_profiles.sh headline 300 r4_prof_ktree_config1 "stats 3 4" --net mini --games 4096 --sims 200
python3 - <<'PY'
import json
for name in ("r4_prof_ktree_headline", "r4_prof_ktree_config1"):
    d = json.load(open(f"gpurun_out/{name}/summary_traffic.json"))
    for k, v in d["kernels"].items():
        if k.startswith("k_tree") or k.startswith("k_net_mfma"):
            print(name, k, v)
PY
