#!/usr/bin/env python
"""Static instruction mix of a kernel's main loop from the compiler's gfx950 assembly (measurement helper, not product):
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off --cuda-device-only -S csrc/raz_engine.hip -o /tmp/e.s
    python tools/count_loop_valu.py /tmp/e.s k_solve_run
prints the VALU / SALU / VMEM / LDS instruction counts of the whole kernel and of its largest backward-branch loop."""
import re
import sys


def main(path, needle):
    cur, body = None, {}
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
            body[cur] = []
        elif cur:
            body[cur].append(line.rstrip("\n"))
            if line.startswith("\t.end_amdhsa_kernel") or line.startswith("\ts_endpgm") and False:
                cur = None
    for name, lines in body.items():
        if needle not in name:
            continue
        end = next((i for i, l in enumerate(lines) if ".Lfunc_end" in l), len(lines))
        lines = lines[:end]
        labels = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r"^(\.LBB\w+):", l)] if m}

        def mix(ls):
            v = [x for x in ls if re.match(r"^\tv_", x)]
            return {"valu": len(v), "valu_64bit_shifts": len([x for x in v if re.match(r"^\tv_(lshlrev_b64|lshrrev_b64|ashrrev_i64)", x)]),
                    "salu": len([x for x in ls if re.match(r"^\ts_", x)]), "vmem": len([x for x in ls if re.match(r"^\t(global|flat|buffer|scratch)_", x)]),
                    "lds": len([x for x in ls if re.match(r"^\tds_", x)]), "waitcnt": len([x for x in ls if re.match(r"^\ts_waitcnt", x)])}
        best = None
        for i, l in enumerate(lines):
            m = re.search(r"\ts_c?branch\w*\s+(\.LBB\w+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                b = lines[labels[m.group(1)]:i]
                if best is None or len(b) > len(best):
                    best = b
        print(name, "whole kernel", mix(lines), "largest loop", mix(best or []))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
