#!/bin/bash
# quick narrow-net timing: stand-alone forward at 4096 / 16384 positions (median and best of 300), default kernel and a named variant
cd ${GRAFT_REPO_ROOT:-/root/repo}
for k in "" "--kernel ${1:-mfma_lean}"; do for n in 4096 16384; do timeout 60 python tools/bench_net.py --net mini --n $n --iters 300 $k 2>/dev/null | python -c "
import json,sys,statistics; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); ms=d['ms_per_forward']; print(d['kernel'], 'n', d['positions'], 'us per forward: median %.2f best %.2f' % (statistics.median(ms)*1000, min(ms)*1000))"; done; done
