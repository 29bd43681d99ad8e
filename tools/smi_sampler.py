#!/usr/bin/env python3
"""Socket power / clock / throttle telemetry at >= 10 Hz while a measurement runs (measurement harness, not product).

    python tools/smi_sampler.py --out gpurun_out/x/smi.jsonl --hz 20 -- tools/probe_conv power 6 8192

Starts the command, samples GPU 0 through the amdsmi Python binding until it exits (one JSON line per sample, UNIX time
stamps), then - if the command printed `{"experiment": "power", "phase": ..., "t_start_unix": ..., "t_end_unix": ...}` lines
(tools/probe_conv power) - prints one summary line per phase: mean / max socket power, the power cap, mean / min gfx clock and
the PPT (package power tracking = power limit) violation activity the firmware reports.  Every field is read defensively: a call
this amdsmi build does not have is recorded once under "errors" and skipped.
"""
import argparse
import json
import subprocess
import sys
import threading
import time


def _num(x):
    try:
        if isinstance(x, (int, float)):
            return x
        return float(str(x).split()[0])
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--hz", type=float, default=20.0)
    ap.add_argument("--gpu", type=int, default=0)
    ap.add_argument("cmd", nargs=argparse.REMAINDER)
    a = ap.parse_args()
    cmd = a.cmd[1:] if a.cmd and a.cmd[0] == "--" else a.cmd
    import amdsmi
    amdsmi.amdsmi_init()
    h = amdsmi.amdsmi_get_processor_handles()[a.gpu]
    errors = {}

    def call(name, *args):
        if name in errors:
            return None
        try:
            return getattr(amdsmi, name)(*args)
        except Exception as e:  # noqa: BLE001
            errors[name] = f"{type(e).__name__}: {e}"
            return None

    static = {"power_cap": call("amdsmi_get_power_cap_info", h), "asic": call("amdsmi_get_gpu_asic_info", h)}
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True)
    lines = []

    def reader():
        for ln in proc.stdout:
            sys.stdout.write(ln)
            sys.stdout.flush()
            lines.append(ln)
    th = threading.Thread(target=reader, daemon=True)
    th.start()
    samples = []
    period = 1.0 / a.hz
    with open(a.out, "w") as f:
        f.write(json.dumps({"static": static}, default=str) + "\n")
        nxt = time.time()
        k = 0
        while proc.poll() is None:
            t = time.time()
            s = {"t": t}
            p = call("amdsmi_get_power_info", h)
            if p:
                s["power"] = {k2: _num(v) for k2, v in p.items()}
            c = call("amdsmi_get_clock_info", h, amdsmi.AmdSmiClkType.GFX)
            if c:
                s["gfx_clk"] = {k2: _num(v) for k2, v in c.items()}
            v = call("amdsmi_get_violation_status", h)
            if v:
                s["violation"] = {k2: (_num(x) if not isinstance(x, (list, tuple)) else None) for k2, x in v.items()
                                  if "ppt" in k2 or "socket_thrm" in k2 or "prochot" in k2 or "hbm_thrm" in k2 or "vr_thrm" in k2 or k2 == "violation_timestamp"}
            if k % int(max(1, a.hz)) == 0:   # the whole metrics table once per second (temperatures, per-XCD clocks, residency accumulators)
                m = call("amdsmi_get_gpu_metrics_info", h)
                if m:
                    keep = ("average_socket_power", "current_socket_power", "current_gfxclks", "current_gfxclk", "average_gfxclk_frequency", "temperature_hotspot",
                            "temperature_mem", "throttle_status", "indep_throttle_status", "accumulated_ppt_residency_acc", "ppt_residency_acc", "prochot_residency_acc",
                            "socket_thm_residency_acc", "vr_thm_residency_acc", "hbm_thm_residency_acc", "accumulation_counter", "gfx_activity", "average_gfx_activity",
                            "gfxclk_lock_status", "firmware_timestamp", "system_clock_counter")
                    s["metrics"] = {k2: m[k2] for k2 in keep if k2 in m}
            samples.append(s)
            f.write(json.dumps(s, default=str) + "\n")
            k += 1
            nxt += period
            d = nxt - time.time()
            if d > 0:
                time.sleep(d)
            else:
                nxt = time.time()
        f.write(json.dumps({"errors": errors}) + "\n")
    th.join(timeout=5)
    phases = []
    for ln in lines:
        try:
            d = json.loads(ln)
        except Exception:
            continue
        if d.get("experiment") == "power":
            phases.append(d)

    def pw(s):
        p = s.get("power") or {}
        for key in ("current_socket_power", "socket_power", "average_socket_power"):
            if p.get(key) not in (None, 0):
                return p[key]
        return None
    cap = None
    if isinstance(static.get("power_cap"), dict):
        cap = _num(static["power_cap"].get("power_cap"))
    for ph in phases:
        ss = [s for s in samples if ph["t_start_unix"] + 0.5 <= s["t"] <= ph["t_end_unix"]]   # (skip the ramp of the first half second)
        P = [pw(s) for s in ss if pw(s) is not None]
        C = [s["gfx_clk"].get("clk") for s in ss if s.get("gfx_clk") and s["gfx_clk"].get("clk") is not None]
        act = [s["violation"].get("active_ppt_pwr") for s in ss if s.get("violation") and s["violation"].get("active_ppt_pwr") is not None]
        per = [s["violation"].get("per_ppt_pwr") for s in ss if s.get("violation") and s["violation"].get("per_ppt_pwr") is not None]
        acc = [s["violation"].get("acc_ppt_pwr") for s in ss if s.get("violation") and s["violation"].get("acc_ppt_pwr") is not None]
        out = {"summary_of_phase": ph["phase"], "positions": ph.get("positions"), "ms_per_launch": ph.get("ms_per_launch"), "samples": len(ss),
               "sample_hz": round(len(ss) / max(1e-9, ph["t_end_unix"] - ph["t_start_unix"] - 0.5), 1),
               "socket_power_w_mean": round(sum(P) / len(P), 1) if P else None, "socket_power_w_max": max(P) if P else None, "power_cap_w": cap,
               "gfx_clk_mhz_mean": round(sum(C) / len(C), 1) if C else None, "gfx_clk_mhz_min": min(C) if C else None, "gfx_clk_mhz_max": max(C) if C else None,
               "ppt_violation_active_share_of_samples": round(sum(1 for x in act if x) / len(act), 3) if act else None,
               "ppt_violation_percent_mean": round(sum(per) / len(per), 1) if per else None,
               "ppt_violation_accumulator_delta": (acc[-1] - acc[0]) if len(acc) > 1 else None}
        print(json.dumps(out))
    if errors:
        print(json.dumps({"smi_calls_unavailable": errors}))
    return proc.returncode


if __name__ == "__main__":
    sys.exit(main())
