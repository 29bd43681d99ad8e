#!/bin/bash
# quick configs[1] steady-state measurement (k_tree / k_net_mfma per-launch times, sims/s): 3 repeats
cd ${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2 3; do
  timeout 90 python bench.py --net mini --games 4096 --sims 200 --steps 600 --warmup 50 --no-extra-legs --no-spotcheck --no-cpu-baseline 2>/tmp/e.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('sims/s %.2fM  step %.2f us  k_tree %.2f us  k_net %.2f us' % (d['value']/1e6, d['ms_per_step']*1000, d['kernels']['k_tree']['avg_ms']*1000, d['roofline']['avg_kernel_ms']*1000))" || tail -3 /tmp/e.txt
done
