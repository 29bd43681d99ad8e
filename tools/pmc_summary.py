"""Average the rocprofv3 PMC passes written by tools/run_profiles.sh per kernel and per dispatch.

usage: python tools/pmc_summary.py gpurun_out/prof_final profiles/r3_pmc/headline [--last K] [--sweep BOARDS]
writes <out>_pmc_per_dispatch.json (counter averages) and, when BOTH traffic passes are present, <out>_traffic.json (HBM bytes
per launch; FETCH_SIZE / WRITE_SIZE are reported in KiB) plus <out>_config3_traffic.json (one net forward; bench.py reads it
for roofline.traffic).  A traffic pass that is missing for a kernel is an ERROR (exit code 2), never a silent zero: round 2
shipped a WRITE-only figure that was below the algorithmic bytes.
--last K: average only the last K dispatches of every kernel (the timed launches of tools/bench_sweep.py come after the
harvest's launches of the same kernels).  --sweep BOARDS: also write <out-dir>/sweep_traffic.json entries "k_step@BOARDS" /
"k_legal_moves@BOARDS" (merged into an existing file), FETCH doubled as MI355X_MICROARCH.md prescribes for wide coalesced reads."""
import argparse
import collections
import csv
import glob
import json
import os
import sys

SHORT = ["k_solve_run", "k_solve_scan", "k_tree_par_net", "k_tree_net", "k_tree_par", "k_tree", "k_net_mfma", "k_conv3x3_wide", "k_heads_wide", "k_conv0_wide", "k_conv3x3_f16x3", "k_conv0_split",
         "k_heads_split", "k_stats", "k_start", "k_gc", "k_step", "k_legal_moves", "k_leaf_claim", "k_leaf_resolve", "k_leaf_fill"]
TRAFFIC = ("FETCH_SIZE", "WRITE_SIZE")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def provenance():
    """What the traffic files say about the build they were measured on: sha256 over the kernel sources (bench.py recomputes it and
    drops a figure whose sources have changed since) and the time of the pass."""
    import hashlib
    import time
    sys.path.insert(0, ROOT)
    from bench import kernel_sources_sha256
    return {"kernel_sources_sha256": kernel_sources_sha256(), "measured_utc": time.strftime("%Y-%m-%d %H:%M:%S", time.gmtime()),
            "measured_by": "tools/pmc_summary.py on the GPU box, from the rocprofv3 --pmc passes of tools/run_profiles.sh"}



def short(name):
    for k in SHORT:
        if k in name:
            return k
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("out")
    ap.add_argument("--last", type=int, default=0)
    ap.add_argument("--sweep", type=int, default=0)
    a = ap.parse_args()
    rows = collections.defaultdict(lambda: collections.defaultdict(dict))   # kernel@grid -> counter -> {dispatch id: value}
    passes = set()
    for path in sorted(glob.glob(os.path.join(a.src, "pmc*", "**", "*counter_collection.csv"), recursive=True)):
        with open(path) as f:
            for r in csv.DictReader(f):
                k = short(r.get("Kernel_Name", ""))
                if not k:
                    continue
                k = f"{k}@grid{r.get('Grid_Size')}"   # slices and whole-batch launches are different workloads
                c = r["Counter_Name"]
                passes.add(c)
                d = rows[k][c]
                did = int(r.get("Dispatch_Id") or 0)
                d[did] = d.get(did, 0.0) + float(r["Counter_Value"])   # (a counter may be split over several rows of one dispatch)
    if not rows:
        print(f"pmc_summary: no counter_collection.csv with known kernels under {a.src}/pmc*", file=sys.stderr)
        return 2
    res = {}
    for k, cs in rows.items():
        res[k] = {}
        for c, d in cs.items():
            ids = sorted(d)
            if a.last:
                ids = ids[-a.last:]
            res[k][c] = sum(d[i] for i in ids) / len(ids)
        res[k]["dispatches"] = max(len(d) for d in cs.values())
    # the most frequent grid of each kernel (the in-run slice launches) also goes under the bare kernel name
    for base in sorted({k.split("@")[0] for k in res}):
        modal = max((k for k in res if k.startswith(base + "@")), key=lambda k: res[k]["dispatches"])
        res[base] = dict(res[modal], grid=modal.split("@grid")[1])
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    with open(a.out + "_pmc_per_dispatch.json", "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    print(json.dumps({k: {c: (round(v, 1) if isinstance(v, float) else v) for c, v in res[k].items()}
                      for k in ("k_solve_run", "k_tree", "k_tree_par", "k_tree_net", "k_tree_par_net", "k_net_mfma", "k_conv3x3_f16x3", "k_conv3x3_wide", "k_step", "k_legal_moves") if k in res}, indent=1))

    have = [c for c in TRAFFIC if c in passes]
    if not have:
        print("pmc_summary: no FETCH_SIZE / WRITE_SIZE pass in this directory: no traffic file written")
        return 0
    missing = sorted({f"{k}: {c}" for k, v in res.items() if "@" not in k for c in TRAFFIC if c not in v and any(t in v for t in TRAFFIC)}
                     | {f"(whole pass) {c}" for c in TRAFFIC if c not in passes})
    if missing:
        print("pmc_summary: ERROR - traffic pass incomplete, refusing to write a traffic figure:\n  " + "\n  ".join(missing), file=sys.stderr)
        return 2
    traffic = {k: {"fetch_bytes_raw": v["FETCH_SIZE"] * 1024.0, "write_bytes": v["WRITE_SIZE"] * 1024.0,
                   # FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at 64 B on gfx950 (MI355X_MICROARCH.md): doubled
                   "hbm_bytes_per_launch": (2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0,
                   "hbm_bytes_per_launch_uncorrected": (v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0, "dispatches_averaged": v["dispatches"] if not a.last else min(a.last, v["dispatches"]),
                   "grid_threads": int(v["grid"]) if str(v.get("grid", "")).isdigit() else None}
               for k, v in res.items() if "@" not in k and all(t in v for t in TRAFFIC)}
    with open(a.out + "_traffic.json", "w") as f:
        json.dump({"source": "separate rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes (tools/run_profiles.sh), KiB -> bytes, average per dispatch; "
                             "hbm_bytes_per_launch = 2 x FETCH + WRITE (FETCH_SIZE under-reports wide coalesced reads by 2 on gfx950, MI355X_MICROARCH.md)",
                   "fetch_pass_present": True, "write_pass_present": True, "kernels": traffic, "provenance": provenance()}, f, indent=1, sort_keys=True)
    # one net forward of the wide nets = conv0 + 2R conv launches + heads: HBM bytes per forward for bench.py's roofline.traffic
    for conv, c0, hd in (("k_conv3x3_f16x3", "k_conv0_split", "k_heads_split"), ("k_conv3x3_wide", "k_conv0_wide", "k_heads_wide")):
        if conv in traffic and c0 in traffic and hd in traffic and res[c0]["dispatches"]:
            per_fwd = res[conv]["dispatches"] / res[c0]["dispatches"]
            with open(a.out + "_config3_traffic.json", "w") as f:
                json.dump({"source": "separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the headline bench command (tools/run_profiles.sh); FETCH doubled "
                                     "(MI355X_MICROARCH.md: FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads on gfx950)",
                           "fetch_pass_present": True, "write_pass_present": True, "provenance": provenance(),
                           "conv_kernel": conv, "conv_launches_per_forward": per_fwd,
                           "conv_fetch_bytes_per_launch_raw": traffic[conv]["fetch_bytes_raw"], "conv_write_bytes_per_launch": traffic[conv]["write_bytes"],
                           "net_forward_hbm_bytes_per_launch": traffic[c0]["hbm_bytes_per_launch"] + per_fwd * traffic[conv]["hbm_bytes_per_launch"]
                           + traffic[hd]["hbm_bytes_per_launch"],
                           "net_forward_hbm_bytes_per_launch_uncorrected": traffic[c0]["hbm_bytes_per_launch_uncorrected"]
                           + per_fwd * traffic[conv]["hbm_bytes_per_launch_uncorrected"] + traffic[hd]["hbm_bytes_per_launch_uncorrected"]}, f, indent=1)
            break
    if a.sweep:
        path = os.path.join(os.path.dirname(os.path.abspath(a.out)), "sweep_traffic.json")
        cur = {}
        if os.path.exists(path):
            with open(path) as f:
                cur = json.load(f)
        for k in ("k_step", "k_legal_moves"):
            if k not in traffic:
                print(f"pmc_summary: ERROR - --sweep given but no traffic for {k}", file=sys.stderr)
                return 2
            cur[f"{k}@{a.sweep}"] = traffic[k]
        with open(path, "w") as f:
            json.dump(cur, f, indent=1, sort_keys=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
