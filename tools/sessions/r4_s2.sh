#!/bin/bash
# Round 4, GPU session 2: k_conv3x3_wino's first hardware run.  Stand-alone timing first (random vs zero operands, tile rounds),
# then the python check: accuracy next to v2 against fp32 / f64 torch, batch invariance, ms per forward.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=$PWD/gpurun_out/r4_s2; mkdir -p $O
timeout 120 tools/probe_wino time 20 > $O/probe_wino.jsonl 2> $O/probe_wino.err; echo "probe_wino rc=$?"; cat $O/probe_wino.jsonl; tail -3 $O/probe_wino.err
timeout 60 tools/probe_conv time 20 > $O/probe_conv.jsonl 2>&1; head -2 $O/probe_conv.jsonl
timeout 500 python tools/sessions/r4_s2.py > $O/wino_check.jsonl 2> $O/wino_check.err; echo "check rc=$?"; tail -1 $O/wino_check.jsonl | head -c 3000; tail -5 $O/wino_check.err
