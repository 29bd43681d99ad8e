"""CPU: the measurement tooling's failure modes.  tools/pmc_summary.py must never turn a missing counter pass into a figure
(round 2 shipped a WRITE-only `roofline.traffic` below the algorithmic bytes because a FETCH_SIZE pass had been dropped)."""
import csv
import json
import os
import subprocess
import sys

from conftest import ROOT


def _pass(root, name, counter, rows):
    d = root / name / "x"
    d.mkdir(parents=True)
    with open(d / "1_counter_collection.csv", "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=["Dispatch_Id", "Kernel_Name", "Grid_Size", "Counter_Name", "Counter_Value"])
        w.writeheader()
        for i, (kernel, value) in enumerate(rows):
            w.writerow(dict(Dispatch_Id=i, Kernel_Name=kernel, Grid_Size=4096, Counter_Name=counter, Counter_Value=value))


def _run(src, out, *extra):
    return subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), str(src), str(out), *extra], capture_output=True, text=True)


def test_pmc_summary_needs_both_traffic_passes(tmp_path):
    rows = [("void (anonymous namespace)::k_step(x)", 100.0), ("void (anonymous namespace)::k_step(x)", 300.0), ("k_legal_moves(y)", 50.0)]
    _pass(tmp_path / "half", "pmc3", "FETCH_SIZE", rows)
    r = _run(tmp_path / "half", tmp_path / "out_half" / "s")
    assert r.returncode == 2 and "refusing" in r.stderr and "WRITE_SIZE" in r.stderr
    assert not (tmp_path / "out_half" / "s_traffic.json").exists()          # no figure, not a zero
    assert (tmp_path / "out_half" / "s_pmc_per_dispatch.json").exists()     # the counters that are there are still summarised
    _pass(tmp_path / "both", "pmc3", "FETCH_SIZE", rows)
    _pass(tmp_path / "both", "pmc4", "WRITE_SIZE", [(k, v / 10) for k, v in rows])
    r = _run(tmp_path / "both", tmp_path / "out_both" / "s", "--last", "1", "--sweep", "4096")
    assert r.returncode == 0, r.stderr
    t = json.load(open(tmp_path / "out_both" / "s_traffic.json"))
    assert t["fetch_pass_present"] and t["write_pass_present"]
    k = t["kernels"]["k_step"]   # --last 1: only the last dispatch; KiB -> bytes; FETCH doubled in hbm_bytes_per_launch
    assert k["fetch_bytes_raw"] == 300 * 1024 and k["write_bytes"] == 30 * 1024 and k["hbm_bytes_per_launch"] == (600 + 30) * 1024
    sw = json.load(open(tmp_path / "out_both" / "sweep_traffic.json"))
    assert set(sw) == {"k_step@4096", "k_legal_moves@4096"}
    # a kernel present in one pass only is an error too
    _pass(tmp_path / "ragged", "pmc3", "FETCH_SIZE", rows)
    _pass(tmp_path / "ragged", "pmc4", "WRITE_SIZE", rows[:2])
    assert _run(tmp_path / "ragged", tmp_path / "out_ragged" / "s").returncode == 2


def test_bench_reports_no_traffic_without_both_passes(tmp_path, monkeypatch):
    """bench.conv_traffic: a committed summary that lacks a pass yields None, never a partial sum."""
    sys.path.insert(0, ROOT)
    import bench
    d = tmp_path / "profiles" / "r3_pmc"
    d.mkdir(parents=True)
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    assert bench.conv_traffic() == (None, None)
    json.dump({"net_forward_hbm_bytes_per_launch": 1.0, "fetch_pass_present": False, "write_pass_present": True}, open(d / "headline_config3_traffic.json", "w"))
    assert bench.conv_traffic() == (None, None)
    json.dump({"net_forward_hbm_bytes_per_launch": 2.5e10, "fetch_pass_present": True, "write_pass_present": True}, open(d / "headline_config3_traffic.json", "w"))
    assert bench.conv_traffic()[0] == 2.5e10


def test_pmc_summary_tells_the_fused_kernels_from_the_classic_ones():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_summary as P
    name = "void (anonymous namespace)::{}<false>(raz_engine_dev, unsigned int, unsigned int)"
    assert P.short(name.format("k_tree")) == "k_tree" and P.short(name.format("k_tree_par")) == "k_tree_par"
    assert P.short(name.format("k_tree_net")) == "k_tree_net" and P.short(name.format("k_tree_par_net")) == "k_tree_par_net"
    assert P.short("(anonymous namespace)::k_conv3x3_f16x3(unsigned char const*, ...)") == "k_conv3x3_f16x3"


def test_bench_child_legs_never_cost_the_parent_its_line():
    """bench.child_leg: the document a child prints, or what went wrong - a child that crashes, prints nothing, or hangs (killed after
    the timeout) yields an error entry, not an exception and not a hung parent."""
    sys.path.insert(0, ROOT)
    import time
    import bench
    assert bench.child_leg(["-c", "print('noise'); print('{\"value\": 3}')"], 30.0) == {"value": 3}
    assert "exit code 7" in bench.child_leg(["-c", "import sys; sys.exit(7)"], 30.0)["error"]
    assert "exit code" in bench.child_leg(["-c", "import os; os.abort()"], 30.0)["error"]
    assert "error" in bench.child_leg(["-c", "print('no json here')"], 30.0)
    t0 = time.time()
    assert "did not finish" in bench.child_leg(["-c", "import time; time.sleep(60)"], 1.0)["error"]
    assert time.time() - t0 < 20
