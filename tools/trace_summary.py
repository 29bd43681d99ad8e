#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace CSV on the GPU box (the trace itself is too large to bring back): per kernel - launches, total
time, duration quantiles - plus the union of busy time and the wall span, for a chosen part of the run.
usage: python tools/trace_summary.py <kernel_trace.csv> [out.json]"""
import csv
import json
import sys


def main(path, out=None):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    res = {"wall_span_ms": (t1 - t0) / 1e6, "launches": len(rows)}
    busy, cur_s, cur_e = 0, None, None
    for s, e, _ in rows:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    res["gpu_busy_union_ms"] = busy / 1e6
    per = {}
    for s, e, n in rows:
        import re
        m = re.search(r"(k_\w+(?:<[\w, ]+>)?)", n)
        short = m.group(1) if m else n[:40]
        per.setdefault(short, []).append(e - s)
    ks = {}
    for n, d in per.items():
        d.sort()
        q = lambda p: d[min(len(d) - 1, int(p * len(d)))] / 1e3
        ks[n] = {"launches": len(d), "total_ms": sum(d) / 1e6, "p10_us": q(0.1), "p50_us": q(0.5), "p90_us": q(0.9), "p99_us": q(0.99), "max_us": d[-1] / 1e3,
                 "launches_over_20us": sum(1 for x in d if x > 20000), "total_ms_of_those": sum(x for x in d if x > 20000) / 1e6}
    res["kernels"] = dict(sorted(ks.items(), key=lambda kv: -kv[1]["total_ms"]))
    # the second half of the run by time (the solver-bound phase of a lock-step whole-game leg)
    txt = json.dumps(res, indent=1)
    if out:
        open(out, "w").write(txt)
    print(txt[:6000])


if __name__ == "__main__":
    main(*sys.argv[1:3])
