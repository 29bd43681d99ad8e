"""Average the rocprofv3 PMC passes written by tools/run_profiles.sh per kernel and per dispatch.

usage: python tools/pmc_summary.py gpurun_out/prof_final profiles/r1_pmc/final
writes <out>_pmc_per_dispatch.json (counter averages) and <out>_traffic.json (HBM bytes per launch,
FETCH_SIZE / WRITE_SIZE are reported in KiB; bench.py reads the latter for roofline.traffic)."""
import collections
import csv
import glob
import json
import os
import sys

SHORT = {"k_tree_par": "k_tree_par", "k_tree": "k_tree", "k_net_mfma": "k_net_mfma", "k_conv3x3_wide": "k_conv3x3_wide", "k_heads_wide": "k_heads_wide",
         "k_conv0_wide": "k_conv0_wide", "k_conv3x3_f16x3": "k_conv3x3_f16x3", "k_conv0_split": "k_conv0_split", "k_heads_split": "k_heads_split",
         "k_stats": "k_stats", "k_start": "k_start", "k_gc": "k_gc"}


def short(name):
    for k in SHORT:
        if k in name:
            return SHORT[k]
    return None


def main(src, out):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.defaultdict(set))
    for path in sorted(glob.glob(os.path.join(src, "pmc*", "**", "*counter_collection.csv"), recursive=True)):
        with open(path) as f:
            for r in csv.DictReader(f):
                k = short(r.get("Kernel_Name", ""))
                if not k:
                    continue
                k = f"{k}@grid{r.get('Grid_Size')}"   # slices and whole-batch launches are different workloads
                c = r["Counter_Name"]
                acc[k][c] += float(r["Counter_Value"])
                cnt[k][c].add(r.get("Dispatch_Id"))
    res = {k: {c: acc[k][c] / max(len(cnt[k][c]), 1) for c in acc[k]} for k in acc}
    for k in res:
        res[k]["dispatches"] = max(len(v) for v in cnt[k].values())
    # the most frequent grid of each kernel (the in-run slice launches) also goes under the bare kernel name
    for base in sorted({k.split("@")[0] for k in res}):
        modal = max((k for k in res if k.startswith(base + "@")), key=lambda k: res[k]["dispatches"])
        res[base] = dict(res[modal], grid=modal.split("@grid")[1])
    with open(out + "_pmc_per_dispatch.json", "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    traffic = {k: {"fetch_bytes": v.get("FETCH_SIZE", 0.0) * 1024.0, "write_bytes": v.get("WRITE_SIZE", 0.0) * 1024.0,
                   "hbm_bytes_per_launch": (v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024.0}
               for k, v in res.items() if "@" not in k and ("FETCH_SIZE" in v or "WRITE_SIZE" in v)}
    with open(out + "_traffic.json", "w") as f:
        json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/run_profiles.sh), KiB -> bytes, "
                             "average per dispatch; FETCH_SIZE under-reports wide coalesced reads on gfx950 (guide)",
                   "kernels": traffic}, f, indent=1, sort_keys=True)
    # one net forward of the wide nets = conv0 + 2R conv launches + heads: HBM bytes per forward for bench.py's roofline.traffic
    for conv, c0, hd in (("k_conv3x3_f16x3", "k_conv0_split", "k_heads_split"), ("k_conv3x3_wide", "k_conv0_wide", "k_heads_wide")):
        if conv in traffic and c0 in traffic and hd in traffic and res[c0]["dispatches"]:
            per_fwd = res[conv]["dispatches"] / res[c0]["dispatches"]
            fwd = traffic[c0]["hbm_bytes_per_launch"] + per_fwd * traffic[conv]["hbm_bytes_per_launch"] + traffic[hd]["hbm_bytes_per_launch"]
            with open(out + "_config3_traffic.json", "w") as f:
                json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the headline bench command (tools/run_profiles.sh); "
                                     "FETCH_SIZE is reported at 1/2 of the bytes of wide coalesced reads on gfx950 (MI355X_MICROARCH.md): "
                                     "fetch bytes doubled here as the guide prescribes",
                           "conv_kernel": conv, "conv_launches_per_forward": per_fwd,
                           "conv_fetch_bytes_per_launch_raw": traffic[conv]["fetch_bytes"], "conv_write_bytes_per_launch": traffic[conv]["write_bytes"],
                           "net_forward_hbm_bytes_per_launch": traffic[c0]["hbm_bytes_per_launch"] + traffic[c0]["fetch_bytes"]
                           + per_fwd * (traffic[conv]["hbm_bytes_per_launch"] + traffic[conv]["fetch_bytes"])
                           + traffic[hd]["hbm_bytes_per_launch"] + traffic[hd]["fetch_bytes"],
                           "net_forward_hbm_bytes_per_launch_uncorrected": fwd}, f, indent=1)
            break
    print(json.dumps({k: {c: (round(v, 1) if isinstance(v, float) else v) for c, v in res[k].items()}
                      for k in ("k_tree", "k_tree_par", "k_net_mfma", "k_conv3x3_f16x3", "k_conv3x3_wide") if k in res}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
