#!/usr/bin/env python
"""tools/count_valu.py — static VALU-issue floor of the sweep kernels from the compiler's gfx950 assembly:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off --cuda-device-only -S csrc/raz_sweep.hip -o sweep.s
    python tools/count_valu.py sweep.s
For k_step / k_legal_moves it finds the main loop (the largest backward-branch loop body of the kernel), counts its vector-ALU
instructions (v_* that are not memory: no global_/flat_/buffer_/ds_/scratch_), and turns that into time: one wave64 VALU
instruction occupies a SIMD's 16 lanes for 4 cycles, a CU has 4 SIMDs, the chip 256 CUs at <= 2.4 GHz, so the chip issues at
most 256 x 4 / 4 = 256 wave-instructions per cycle = 16384 lane-operations per cycle.  The loop body of both kernels handles
4 boards per lane = 256 boards per wave."""
import re
import sys


def kernels(path):
    cur, out = None, {}
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
            out[cur] = []
        elif cur and line.startswith("\t.end_amdhsa_kernel") is False and cur is not None:
            out[cur].append(line.rstrip("\n"))
    return out


def main(path):
    CLK, CUS = 2.4e9, 256
    for name, lines in kernels(path).items():
        short = ("k_step_sliced" if "k_step_sliced" in name else "k_step_hybrid" if "k_step_hybrid" in name else "k_legal_moves_sliced" if "k_legal_moves_sliced" in name else
                 "k_step" if "6k_step" in name else ("k_legal_moves" if "k_legal_moves" in name else None))
        if not short:
            continue
        if short.endswith("_sliced") or short == "k_step_hybrid":   # straight-line kernels: one superblock of 2048 boards (32 per lane) per loop iteration; AGPR copies are VALU work too
            valu = [x for x in lines if re.match(r"^\tv_", x)]
            moves = [x for x in valu if "accvgpr" in x or re.match(r"^\tv_mov_b32", x)]
            print(f"{short}: {len(valu)} VALU instructions per wave iteration of 2048 boards (32 boards per lane, bit-sliced) = {len(valu) / 32.0:.1f} per board, "
                  f"{len(moves)} of them register copies (v_accvgpr_read / write, v_mov: {len(moves) / 32.0:.1f} per board); the whole kernel incl. its rare paths.  "
                  f"VALU-issue floor at 2^24 boards: {(1 << 24) * len(valu) / 32.0 / (CUS * 64.0) / CLK * 1e3:.4f} ms")
            hist = {}
            for x in valu:
                hist[x.split()[0]] = hist.get(x.split()[0], 0) + 1
            print("   ", ", ".join(f"{k} x{v}" for k, v in sorted(hist.items(), key=lambda kv: -kv[1])[:14]))
            continue
        labels = {}
        for i, l in enumerate(lines):
            m = re.match(r"^(\.LBB\w+):", l)
            if m:
                labels[m.group(1)] = i
        best = None
        for i, l in enumerate(lines):   # backward branches = loops; the main loop is the one with the most VALU inside
            m = re.search(r"\ts_c?branch\w*\s+(\.LBB\w+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                body = lines[labels[m.group(1)]:i]
                valu = [x for x in body if re.match(r"^\tv_", x)]
                if best is None or len(valu) > len(best[0]):
                    best = (valu, body)
        valu, body = best
        mem = [x for x in body if re.match(r"^\t(global|flat|buffer|scratch)_", x)]
        salu = [x for x in body if re.match(r"^\ts_", x)]
        wide = [x for x in valu if re.match(r"^\tv_(lshlrev_b64|lshrrev_b64|ashrrev_i64)", x)]
        per_board = len(valu) / 4.0
        boards = 1 << 24
        t1 = boards * per_board / (CUS * 64.0) / CLK                         # every instruction one 4-cycle pass
        t2 = boards * (len(valu) + len(wide)) / 4.0 / (CUS * 64.0) / CLK     # 64-bit shifts two passes (half rate)
        print(f"{short}: main loop = {len(valu)} VALU + {len(salu)} SALU + {len(mem)} VMEM instructions per wave iteration of 256 boards "
              f"(4 boards per lane) = {per_board:.1f} VALU instructions per board, {len(wide)} of the {len(valu)} being 64-bit shifts.  "
              f"VALU-issue floor at 2^24 boards ({CUS} CUs x 4 SIMDs x 16 lanes, {CLK / 1e9:.1f} GHz): {t1 * 1e3:.4f} ms if every instruction is "
              f"one 4-cycle pass, {t2 * 1e3:.4f} ms with the 64-bit shifts at half rate")
        hist = {}
        for x in valu:
            op = x.split()[0]
            hist[op] = hist.get(op, 0) + 1
        print("   ", ", ".join(f"{k} x{v}" for k, v in sorted(hist.items(), key=lambda kv: -kv[1])[:14]))


if __name__ == "__main__":
    main(sys.argv[1])
