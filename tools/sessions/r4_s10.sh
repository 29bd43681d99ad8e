#!/bin/bash
# Round 4, GPU session 13: rocprofv3 kernel statistics + counter passes of the headline command and of the configs[1] batch on the
# fused tree + net kernel (tools/run_profiles.sh: one counter set per pass, --kernel-trace only).
cd ${GRAFT_REPO_ROOT:-/root/repo}
PROF_TIMEOUT=400 bash tools/run_profiles.sh headline 20 r4_prof_headline "stats 3 4"
PROF_TIMEOUT=300 bash tools/run_profiles.sh headline 40 r4_prof_config1_fused "stats 1 2 3 4" --net mini --games 4096 --sims 200 --fused
ls -la gpurun_out/r4_prof_headline gpurun_out/r4_prof_config1_fused | head -40
cat gpurun_out/r4_prof_headline/summary_pmc.txt | tail -5; cat gpurun_out/r4_prof_config1_fused/summary_pmc.txt | tail -5
