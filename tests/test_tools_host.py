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


def test_bench_reports_no_traffic_without_both_passes_or_from_other_sources(tmp_path, monkeypatch):
    """bench.conv_traffic: a committed summary that lacks a pass yields None, never a partial sum - and so does one that was measured
    on other kernel sources than the ones this run is built from (the file records their sha256)."""
    sys.path.insert(0, ROOT)
    import bench
    d = tmp_path / "profiles" / bench.PMC_DIR
    d.mkdir(parents=True)
    csrc = tmp_path / "reversi-alpha-zero_amd" / "csrc"
    csrc.mkdir(parents=True)
    (tmp_path / "include").mkdir()
    (csrc / "k.hip").write_text("kernel v1")
    (tmp_path / "include" / "raz.h").write_text("abi")
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    v, why = bench.conv_traffic()
    assert v is None and "no counter pass" in why
    sha = bench.kernel_sources_sha256()
    prov = {"kernel_sources_sha256": sha, "measured_utc": "2026-09-25 20:00:00"}
    json.dump({"net_forward_hbm_bytes_per_launch": 1.0, "fetch_pass_present": False, "write_pass_present": True, "provenance": prov}, open(d / "headline_config3_traffic.json", "w"))
    assert bench.conv_traffic()[0] is None
    json.dump({"net_forward_hbm_bytes_per_launch": 2.5e10, "fetch_pass_present": True, "write_pass_present": True, "provenance": prov}, open(d / "headline_config3_traffic.json", "w"))
    v, src = bench.conv_traffic()
    assert v == 2.5e10 and "2026-09-25 20:00:00" in src and sha[:12] in src
    (csrc / "k.hip").write_text("kernel v2")   # the kernels changed after the counter pass: the figure is withheld, and the line says why
    v, why = bench.conv_traffic()
    assert v is None and "OTHER kernel sources" in why
    json.dump({"net_forward_hbm_bytes_per_launch": 2.5e10, "fetch_pass_present": True, "write_pass_present": True}, open(d / "headline_config3_traffic.json", "w"))
    assert bench.conv_traffic()[0] is None   # no provenance at all (a file of an earlier round)


def test_pmc_summary_tells_the_fused_kernels_from_the_classic_ones():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_summary as P
    name = "void (anonymous namespace)::{}<false>(raz_engine_dev, unsigned int, unsigned int)"
    assert P.short(name.format("k_tree")) == "k_tree" and P.short(name.format("k_tree_par")) == "k_tree_par"
    assert P.short(name.format("k_tree_net")) == "k_tree_net" and P.short(name.format("k_tree_par_net")) == "k_tree_par_net"
    assert P.short("(anonymous namespace)::k_conv3x3_f16x3(unsigned char const*, ...)") == "k_conv3x3_f16x3"


def test_bench_line_is_compact_and_keeps_the_contract():
    """bench.compact_line: whatever the legs put into the full document, the printed line stays below 8 KB and carries the contract
    keys, `roofline`, `cpu_baseline`, every leg's value and every parity result (the driver's record keeps what is in the line)."""
    sys.path.insert(0, ROOT)
    import bench
    blah = "x" * 5000
    games = [{"game_id": i, "sims": 800, "note": blah} for i in range(8)]
    full = {"metric": "MCTS simulations/sec (self-play, NN included)", "value": 3.5e5, "unit": "sims/s", "n_gpus": 1, "steps": 20, "warmup": 5,
            "ms_per_step": 25.0, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64 + f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2] ...", "games_per_gpu": 8192, "sims_per_move": 800, "net": "ch5", "slices": 1, "junk": blah},
            "roofline": {"bound": "mfma", "kernel": blah, "kernel_name": "k_conv3x3_f16x3", "achieved": 450.0, "peak": 2500.0, "unit": "TFLOP/s", "frac": 0.18,
                         "traffic": 2.8e10, "peak_note": blah},
            "cpu_baseline": {"value": 900.0, "unit": "sims/s", "cores": 256, "kind": "reference", "sample": "20 s window", "detail": blah,
                             "configs1_mini_200sims": {"value": 28000.0}},
            "games_per_hour": 27000.0, "leaves_per_sec": 3.0e5,
            "parity_spotcheck": {"result": "ok", "what": blah, "games": games}, "parity_spotcheck_timed_batch": {"result": "ok", "what": blah, "games": games},
            "whole_games_measured": {"value": 3.4e5, "unit": "sims/s", "seconds": 120.0, "games_per_hour": 26500.0, "workload": blah,
                                     "parity_check_complete_games": {"result": "ok", "what": blah, "games": games[:2]}},
            "ch5_yml_as_shipped": {"value": 3.0e5, "unit": "sims/s", "workload": blah, "same_with_the_solver_off": {"value": 3.1e5},
                                   "solver_share_of_a_step": {"share_of_step_time": 0.03}},
            "config1_4096x200_mini": {"value": 1.0e8, "unit": "sims/s", "games_per_hour": 3.0e7, "workload": blah, "fused_tree_net_kernel": True,
                                      "parity_spotcheck": {"result": "ok", "games": games}},
            "config5_8192x3200_agz": {"error": "RuntimeError('x')"},
            "bitboard_sweep": {"k_step": {"achieved": 5000.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.63, "traffic_over_algorithmic": 1.0003, "note": blah},
                               "k_legal_moves": {"achieved": 5600.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.70},
                               "beyond_the_infinity_cache_2^26_boards": {"k_step": {"frac": 0.63}, "k_legal_moves": {"frac": 0.71}}},
            "record_gather": {"collective": "none (1 GPU)", "bytes": 0, "seconds": 0.0}, "_full_path": "gpurun_out/bench_full.json"}
    line = bench.compact_line(full)
    text = json.dumps(line)
    assert len(text) < 8192, len(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert line[k] == full[k]
    assert line["config"]["workload"] == full["config"]["workload"] and "junk" not in line["config"]
    assert line["roofline"]["frac"] == 0.18 and line["roofline"]["kernel_name"] == "k_conv3x3_f16x3" and "kernel" not in line["roofline"]
    assert line["cpu_baseline"] == {"value": 900.0, "unit": "sims/s", "cores": 256, "kind": "reference", "sample": "20 s window", "configs1_value": 28000.0}
    assert line["parity_spotcheck"] == {"result": "ok", "games": 8} and line["whole_games_measured"]["parity"] == {"result": "ok", "games": 2}
    assert line["whole_games_measured"]["sims_per_s"] == 3.4e5 and line["whole_games_measured"]["games_per_hour"] == 26500.0
    assert line["ch5_yml_as_shipped"]["solver_share_of_step_time"] == 0.03 and line["ch5_yml_as_shipped"]["solver_off_value"] == 3.1e5
    assert line["config1_4096x200_mini"]["value"] == 1.0e8 and line["config1_4096x200_mini"]["parity"] == {"result": "ok", "games": 8}
    assert line["config5_8192x3200_agz"] == {"error": "RuntimeError('x')"} and line["bitboard_sweep"]["k_step"]["frac"] == 0.63
    assert line["full_document"] == "gpurun_out/bench_full.json"



def test_train_then_check_v2_runs_end_to_end_on_a_small_net(tmp_path):
    """tools/train_then_check_v2.py (the evidence file profiles/r5/net_v2_on_a_cpu_trained_256x10_net_wave_emulator.json) at a size
    a test can afford: oracle self-play rows, a few SGD steps of a 128-filter net, then the product's v1 and v2 kernel sources on
    the wave emulator against fp32 torch - within the path's tolerance, range flag clear, weights really moved."""
    out = tmp_path / "acc.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "train_then_check_v2.py"), "--games", "24", "--sims", "8", "--steps", "12",
                        "--net", "128,1,32", "--emu-positions", "3", "--torch-positions", "64", "--threads", "2", "--out", str(out)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    d = json.load(open(out))
    e = d["emulated_kernels"]
    assert d["within_tolerance"] is True and e["v2_range_flag"] is False and e["positions"] == 3
    assert max(e["v1_vs_fp32_torch"]["policy"]["max"], e["v1_vs_fp32_torch"]["value"]["max"]) <= 1e-5
    assert d["what_training_changed"]["weights_moved_from_init_rel_l2"] > 0 and d["training"]["rows"] > 200
