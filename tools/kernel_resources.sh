#!/bin/bash
# Compiler-reported resources of the engine's kernels (VGPRs, spills, scratch, occupancy): hipcc -Rpass-analysis=kernel-resource-usage
# over the two engine translation units, one line per kernel.  usage: tools/kernel_resources.sh [out.txt]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-/dev/stdout}
cd "$ROOT/reversi-alpha-zero_amd/csrc"
for f in raz_engine.hip raz_engine_fused.hip raz_sweep.hip; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -c $f -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 | \
  python3 -c "
import re, sys
cur = None; rows = {}
for ln in sys.stdin:
    m = re.search(r'remark: Function Name: (\S+)', ln)
    if m: cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r'remark:\s+(.+?): (\S+) \[-Rpass', ln)
    if m and cur: rows[cur][m.group(1).strip()] = m.group(2)
import subprocess
for k, v in rows.items():
    name = subprocess.run(['c++filt', k], capture_output=True, text=True).stdout.strip()
    mm = re.search(r'(k_\w+(?:<\w+>)?)', name); name = mm.group(1) if mm else name
    print(f\"{name:60s} VGPRs {v.get('VGPRs','?'):>4} AGPRs {v.get('AGPRs','?'):>3} SGPRs {v.get('TotalSGPRs','?'):>4}  VGPR spill {v.get('VGPRs Spill', v.get('VGPR Spill','?')):>4}  SGPR spill {v.get('SGPRs Spill', v.get('SGPR Spill','?')):>4}  scratch B/lane {v.get('ScratchSize [bytes/lane]','?'):>5}  occupancy {v.get('Occupancy [waves/SIMD]','?'):>2}  LDS {v.get('LDS Size [bytes/block]','?')}\")
"
done > "$OUT"
