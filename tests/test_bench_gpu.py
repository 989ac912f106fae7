"""bench.py keeps its contract: one JSON line with the driver's fields, the roofline block of
the integrate kernel and the CPU baseline (with the full-size parity check inside it)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


LINE_BYTES = 6000      # the one stdout line: the driver's fields, roofline, cpu_baseline (flat) and <= 10 keys more


def merged(line):
    """the line plus what it moved to bench_extra.json (bench.py split_line): the object the assertions below read.  Nested
    tables of roofline / cpu_baseline come back under their old keys."""
    j = dict(line)
    ex = line.get("extra")
    if not ex:
        return j
    assert "error" not in ex, ex
    extra = json.load(open(ex["file"]))
    assert sorted(extra) == ex["keys"]
    for k, v in extra.items():
        if k == "cpu_baseline":
            j["cpu_baseline"] = dict(j.get("cpu_baseline") or {}, **v)
        elif k == "roofline_box":
            j["roofline"] = dict(j["roofline"], box=v)
        elif k == "roofline_tables":
            j["roofline"] = dict(j["roofline"], **v)
        else:
            j[k] = v                     # (default_display_frame / post_pass: the whole table replaces the compact one)
    return j


def run_bench(*args, env=None, tmp_extra=None):
    extra_file = os.path.join(tmp_extra or __import__("tempfile").mkdtemp(), "bench_extra.json")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, RGBDR_BENCH_EXTRA=extra_file, **(env or {})))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    if "roofline" in line:
        assert len(lines[0]) < LINE_BYTES, len(lines[0])
        assert len([k for k in line if k not in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                                 "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")]) <= 10, sorted(line)
        assert all(not isinstance(v, (dict, list)) for v in line["roofline"].values()), "roofline is flat on the line"
        if "cpu_baseline" in line:
            assert all(not isinstance(v, (dict, list)) for v in line["cpu_baseline"].values()), "cpu_baseline is flat on the line"
    return merged(line)


def test_bench_line_contract():
    j = run_bench("--steps", "6", "--warmup", "2", "--cpu-rows", "64")
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("dtype", str), ("data", str)):
        assert isinstance(j[key], typ), key
    assert j["n_gpus"] == 1 and j["steps"] == 6 and j["warmup"] == 2 and j["higher_is_better"] is True
    assert j["vs_baseline"] is None and j["scaling"] is None and j["dtype"] == "f32" and j["data"] == "synthetic"
    assert j["unit"] == "Mvoxels/s" and "workload" in j["config"] and "model" not in j["config"]
    assert abs(j["value"] - 512 ** 3 / (j["ms_per_step"] * 1e-3) / 1e6) < 0.01 * j["value"]
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0.5 < r["frac"] < 1.0
    assert abs(r["achieved"] - r["bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 0.01 * r["achieved"]
    assert r["bytes_per_launch"] == 512 ** 3 * (4 + 12 * 4) + 4 * 512 * 424 * 8
    assert r["traffic"] is None or 0.9 * r["bytes_per_launch"] < r["traffic"] < 1.2 * r["bytes_per_launch"]
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "Mvoxels/s" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert c["parity_rows_bit_exact"] is True and c["parity_rows"] == 64 and c["repetitions"] >= 5 and "median of 5" in c["sample"]
    # the reference's own shader text, compiled (it travels as oracle/_ref/libref_shaders.so where the build container had
    # /root/reference): timed on a bounded sample and bit-identical to the HIP rows / images at the benchmark size
    rt = c["reference_shader_text"]
    if rt is not None:
        assert "error" not in rt, rt
        assert rt["hip_rows_bit_identical"] is True and rt["hip_images_bit_identical"] is True and rt["integrate_mvoxels_per_s"] > 0
    # the HIP path against what the reference's own GLSL produced on Mesa (committed fixture): in the bench record
    rg = c["reference_glsl_on_mesa"]
    assert rg is not None and "error" not in rg, rg
    assert rg["tsdf_max_abs_diff"] <= 5e-7 and rg["tsdf_voxels_changing_class"] == 0 and rg["brick_counters_equal"] is True
    assert rg["max_abs_diff"]["depth_rg"] == 0.0 and rg["max_abs_diff"]["depth_b"] == 0.0 and rg["max_abs_diff"]["sil"] == 0.0
    big = rg["baseline_sensor_size"]
    assert big is not None and "error" not in big, big
    assert big["tsdf_max_abs_diff"] <= 5e-7 and big["tsdf_voxels_beyond_1e-6"] == 0 and big["max_abs_diff"]["depth_rg"] == 0.0
    assert big["brick_counts_differing"] <= 1e-3 * big["brick_counts"]
    # box calibration: a replay of the kernel's memory streams on the same box, and the GPU state beside it
    assert r["box_stream_GBps"] > 3000 and abs(r["frac_of_box_stream"] - r["achieved"] / r["box_stream_GBps"]) < 2e-3
    assert 0.8 < r["frac_of_box_stream"] < 1.1
    # clocks / power UNDER LOAD: a burst of at least a second of the same step loop, sampled from a side thread
    box = r["box"]
    assert "error" not in box, box
    if "unavailable" not in box:
        assert box["burst_s"] >= 1.0 and box["samples"] >= 5
        assert box["sclk_MHz"] is None or box["sclk_MHz"]["max"] >= 500.0          # an idle clock is dropped, never printed
        assert box["pci"] is not None and box["power_W"]["median"] > 100.0
    # the placement probe is bench.py's opt-in: the line also carries what the first placement (library default) gives
    assert 0.5 < r["frac_first_placement"] <= r["frac"] * 1.02 and r["avg_launch_ms_first_placement"] > 0
    # ... and what the library's own default (the best of three bounded candidates) gives
    assert r["frac_first_placement"] <= r["frac_library_default"] * 1.001 and r["frac_library_default"] <= r["frac"] * 1.02
    assert r["traffic"] is None or "not this run" in r["traffic_source"]
    assert "4 sensors" in j["metric"] and "512^3" in j["metric"] and j["config"]["baseline_config"].startswith("configs[2]")
    for extra in ("post_pass", "host_fed", "reference_defaults", "bricked", "other_schedule", "full_sweep_store_elision",
                  "full_sweep_background_skip"):
        assert extra in j and "error" not in j[extra], (extra, j[extra])
    assert j["post_pass"]["raymarch_ms"] > 0 and j["post_pass"]["holefill_ms"] > 0 and j["post_pass"]["brickdraw_ms"] > 0
    assert j["bricked"]["ms_per_step"] < j["ms_per_step"]
    assert j["full_sweep_store_elision"]["ms_per_step"] < 1.02 * j["ms_per_step"]
    sk = j["full_sweep_background_skip"]
    assert sk["ms_per_step"] < 0.8 * j["ms_per_step"] and 0.3 < sk["frac_decided"] <= 1.0
    assert sum(sk["verdicts"].values()) == sk["pairs"] and sk["verdicts"]["none"] == sk["pairs"] - sk["pairs_decided"]
    assert sk["GBps"] < 8000.0   # the bytes it asks for over its time: never above the HBM peak
    # the data-dependent modes on more than their best case: static / moving / dense / dense + moving
    sc = j["scenes"]
    assert "error" not in sc, sc
    # the headline's roofline arithmetic per scene, on the line itself (flat, so that the driver's record keeps it)
    for name in ("static", "moving", "dense", "dense_moving"):
        assert r["frac_scene_" + name] == sc[name]["full_sweep"]["roofline_frac"] and 0.5 < r["frac_scene_" + name] < 1.0
        assert 0.9 < r["frac_of_box_stream_scene_" + name] < 1.1
    assert abs(r["frac_scene_static"] - r["frac"]) < 0.08 and r["frac_scene_dense"] > 0.9 * r["frac_scene_static"]     # (a 6-step headline on cold clocks)
    c0 = j["cpu_baseline"]
    assert c0["glsl_on_mesa_tsdf_max_abs_diff"] <= 5e-7 and c0["glsl_on_mesa_brick_counters_equal"] is True
    assert 1e-5 < c0["driver_weight_bound_tsdf_p99_abs_diff_in_band"] < 1e-3        # the derived 8-bit-weight bound
    dd = j["default_display_frame"]
    assert "error" not in dd, dd
    for g in ("reference_box", "grid_512"):
        st = dd[g]["stages_ms"]
        assert 0.1 < dd[g]["ms_per_frame"] < 2.0 and dd[g]["ms_per_frame_moving"] > 0 and 0 < st["holefill"] < 0.09 and st["raymarch"] > 0
        assert st["drawF"] >= st["raymarch"] + st["holefill"] and dd[g]["ms_per_frame"] > st["drawF"]
    assert sc["static"]["valid_pixels"] < 0.5 and sc["dense"]["valid_pixels"] == 1.0 and sc["dense_moving"]["valid_pixels"] > 0.98
    assert sc["moving"]["frames_in_rotation"] == 4 and sc["dense_moving"]["frames_in_rotation"] == 4
    for name in ("static", "moving", "dense", "dense_moving"):
        m = sc[name]
        assert m["pre_chain_ms"] > 0 and m["bricked"]["ms_per_step"] > 0 and 0.0 <= m["background_skip"]["frac_decided"] <= 1.0
        assert abs(m["full_sweep"]["integrate_ms"] - r["avg_launch_ms"]) < 0.25 * r["avg_launch_ms"]      # the headline does not depend on the data
    assert sc["dense"]["pre_chain_ms"] > sc["static"]["pre_chain_ms"]          # 169 taps for every pixel instead of a third of them
    assert sc["dense"]["bricked"]["occupied_ratio"] > sc["static"]["bricked"]["occupied_ratio"]
    assert abs(sc["static"]["bricked"]["ms_per_step"] - j["bricked"]["ms_per_step"]) < 0.3 * j["bricked"]["ms_per_step"]
    rd = j["reference_defaults"]
    assert rd["ms_per_frame_moving"] > 0
    il = j["inverse_lut"]
    assert "error" not in il and il["inverse_lut_generate_ms"] > 0 and il["Gvoxels_per_s"] > 0


def test_a_failing_leg_costs_its_own_key_and_nothing_else():
    """VERDICT r4 task 2: every leg after the headline is isolated; here the brick-skipping leg throws"""
    j = run_bench("--steps", "4", "--warmup", "1", "--cpu-rows", "8", env={"RGBDR_BENCH_FAIL_LEG": "bricked"})
    assert j["bricked"] == {"error": "RuntimeError: RGBDR_BENCH_FAIL_LEG=bricked"}
    assert j["value"] > 0 and 0.5 < j["roofline"]["frac"] < 1.0 and j["cpu_baseline"]["value"] > 0 and j["cpu_baseline"]["parity_rows_bit_exact"] is True
    for later in ("other_schedule", "full_sweep_store_elision", "full_sweep_background_skip", "scenes", "post_pass", "host_fed"):
        assert "error" not in j[later], (later, j[later])


def test_a_hanging_leg_is_ended_by_the_watchdog_and_the_line_is_printed():
    """... and here the host-fed leg never returns: the watchdog prints the line as far as it got, status 0"""
    j = run_bench("--steps", "4", "--warmup", "1", "--no-cpu-baseline", env={"RGBDR_BENCH_FAIL_LEG": "host_fed:hang", "RGBDR_BENCH_LEG_BUDGET": "25"})
    assert "watchdog" in j["host_fed"]["error"] and "host_fed" in j["legs_incomplete"]
    assert j["value"] > 0 and 0.5 < j["roofline"]["frac"] < 1.0 and "error" not in j["bricked"] and "error" not in j["scenes"]


def test_bench_loopback_runs_the_multi_gpu_path():
    j = run_bench("--loopback", "--steps", "6", "--warmup", "2")
    assert j["config"]["halo_transport"] == "rccl" and "loopback" in j["config"]["parallelism"]
    assert j["halo"]["bytes_per_face"] > 0 and j["halo"]["transfer_ms_rank0"] > 0
    assert "cpu_baseline" not in j and j["other_schedule"] is None
    # the slab post-pass of the multi-GPU runs (find, all-reduce MIN, shade, composite, hole filling)
    assert "error" not in j["post_pass"], j["post_pass"]
    assert j["post_pass"]["slab_raymarch_composited_ms"] > 0 and j["post_pass"]["holefill_ms"] > 0


def test_plain_command_line_with_two_gpus_spawns_its_own_ranks():
    """`python3 bench.py --gpus 2` exactly as the driver starts it (no torch.distributed.run around it); gloo because
    the box has one GPU: both ranks share it and the halos travel through the host.  The headline of an N > 1 run is the
    N = 1 workload at fixed work per GPU (4 sensors, 512^3 voxels per rank: value(N) compares with N x value(1)); BASELINE's
    own multi-GPU config is timed in the same run and reported under baseline_configs_run."""
    j = run_bench("--gpus", "2", "--backend", "gloo", "--steps", "3", "--warmup", "1")
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["scaling"] == "weak"
    assert j["launch"]["rung"] == 0 and j["launch"]["line"] == "final" and j["launch"]["failed_attempts"] == [] and j["launch"]["child_status"] == 0
    assert j["config"]["baseline_config"].startswith("configs[2] at fixed work per GPU") and "4 sensors" in j["metric"]
    assert j["config"]["grid"] == [512, 512, 1024] and j["config"]["sensors"] == 4
    pr = j["per_rank"]
    for key in ("integrate_ms", "halo_ms", "ms_per_step", "roofline_frac"):
        assert len(pr[key]) == 2 and all(v is not None and v > 0 for v in pr[key]), (key, pr[key])
    slow = pr["slowest_rank"]
    assert pr["integrate_ms"][slow] == max(pr["integrate_ms"])
    r = j["roofline"]
    assert r["rank"] == slow and abs(r["avg_launch_ms"] - pr["integrate_ms"][slow]) < 1e-3
    assert r["bytes_per_launch"] == 512 ** 3 * (4 + 12 * 4) + 4 * 512 * 424 * 8            # a rank's slab: the N = 1 launch
    assert abs(j["value"] - 512 * 512 * 1024 / (j["ms_per_step"] * 1e-3) / 1e6) < 0.01 * j["value"]
    assert j["ms_per_step"] >= max(pr["ms_per_step"]) * 0.999
    assert "host-staged" in j["config"]["halo_transport"] and j["halo"]["transfer_ms_max"] > 0
    assert "N x value(1)" in j["scaling_note"]
    b = j["baseline_configs_run"]
    assert "error" not in b, b
    assert b["baseline_config"].startswith("configs[3]") and b["grid"] == [512, 512, 512] and b["sensors"] == 8 and b["scaling"] == "strong"
    assert b["value"] > 0 and len(b["per_rank"]["integrate_ms"]) == 2
    assert abs(b["value"] - 512 ** 3 / (b["ms_per_step"] * 1e-3) / 1e6) < 0.01 * b["value"]
    assert abs(b["voxel_sensor_updates_per_s"] - 8 * b["value"] * 1e6) < 0.01 * b["voxel_sensor_updates_per_s"]


def test_two_ranks_started_by_torch_distributed_run():
    """the driver's other way of starting an N > 1 run: `python -m torch.distributed.run ... bench.py --gpus 2`.  Each process
    it starts is a SUPERVISOR (never touches the GPU) whose child does the work on a rendezvous of the children's own; rank
    0's supervisor prints the one line."""
    port = 29500 + os.getpid() % 400
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "3",
                        "--warmup", "1", "--weak"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = merged(json.loads(lines[0]))          # (bench_extra_n2.json beside bench.py: the per-rank tables)
    assert j["n_gpus"] == 2 and j["value"] > 0 and j["scaling"] == "weak" and len(j["per_rank"]["integrate_ms"]) == 2
    assert j["roofline"]["frac_slowest_rank"] == min(j["per_rank"]["roofline_frac"]) and j["roofline"]["ranks"] == 2
    assert j["launch"]["launched_by"] == "torch.distributed.run" and j["launch"]["rung"] == 0 and j["launch"]["line"] == "final"


def test_baseline_configs_as_the_headline_of_a_two_gpu_run():
    """--baseline-configs: configs[3] (8 sensors, 512^3 over the ranks) is the headline, the fixed-work twin the extra key"""
    j = run_bench("--gpus", "2", "--backend", "gloo", "--steps", "3", "--warmup", "1", "--baseline-configs")
    assert j["n_gpus"] == 2 and j["scaling"] == "strong"
    assert j["config"]["baseline_config"].startswith("configs[3]") and "8 sensors" in j["metric"]
    assert j["roofline"]["bytes_per_launch"] == 512 * 512 * 256 * (4 + 12 * 8) + 8 * 512 * 424 * 8
    assert abs(j["value"] - 512 ** 3 / (j["ms_per_step"] * 1e-3) / 1e6) < 0.01 * j["value"]
    w = j["weak_scaling_4_sensors"]
    assert "error" not in w, w
    assert w["grid"] == [512, 512, 1024] and w["sensors"] == 4 and w["value"] > 0 and len(w["per_rank"]["integrate_ms"]) == 2
    assert abs(w["value"] - 512 * 512 * 1024 / (w["ms_per_step"] * 1e-3) / 1e6) < 0.01 * w["value"]
    assert "voxel_sensor_updates_per_s" in j["scaling_note"]


def test_ladder_on_the_gpu_a_hung_first_rung_is_replaced_by_fresh_children():
    """VERDICT r4 task 1 on the real child: rung 0's ranks hang in their init phase (hook), the supervisors end them when the
    rung's budget is up and start rung 1 (--torch-collectives) with fresh processes, which produces the line; two ranks share
    the one GPU, so gloo carries the halos"""
    j = run_bench("--gpus", "2", "--backend", "gloo", "--steps", "3", "--warmup", "1", "--weak", "--rung-budgets", "30,500,500",
                  env={"RGBDR_BENCH_HANG": "0:init", "RGBDR_BENCH_CHAIN": "sharded"})
    la = j["launch"]
    assert la["rung"] == 1 and la["rung_flags"] == ["--torch-collectives", "--no-lagged"] and la["line"] == "final"
    assert len(la["failed_attempts"]) == 1 and la["failed_attempts"][0]["rung"] == 0 and "budget" in la["failed_attempts"][0]["outcome"]
    assert j["n_gpus"] == 2 and j["value"] > 0 and j["config"]["collectives"] == "torch.distributed"
    assert j["config"]["pre_chain"].startswith("sharded by sensor")
    ch = j["config"]["pre_chain_choice"]       # both schedules were timed on the run's own ranks before the headline
    assert ch["kept"] == "sharded" and ch["ms_per_step_sharded"] > 0 and ch["ms_per_step_redundant"] > 0


def test_ladder_last_rung_is_the_redundant_chain_and_the_weak_run_only():
    j = run_bench("--gpus", "2", "--backend", "gloo", "--steps", "3", "--warmup", "1", "--first-rung", "2")
    assert j["launch"]["rung"] == 2 and j["launch"]["rung_flags"] == ["--torch-collectives", "--no-shard", "--no-lagged", "--weak"]
    assert j["config"]["pre_chain"] == "every sensor on every rank" and j["config"]["collectives"] == "torch.distributed"
    assert j["scaling"] == "weak" and "baseline_configs_run" not in j and j["value"] > 0 and len(j["per_rank"]["integrate_ms"]) == 2
    assert "error" not in j["post_pass"] and "error" not in j["bricked"] and "error" not in j["halo"]


def test_the_watchdog_of_a_child_ends_a_phase_that_overruns():
    """a child whose phase exceeds its budget leaves with os._exit (status 75) without waiting for the supervisor: the trial
    step of the library-managed exchange, hung by the hook, with every budget scaled down"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--slab", "1/4", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, RGBDR_BENCH_HANG="-:trial step", RGBDR_BENCH_RUNG="-", RGBDR_BENCH_BUDGET_SCALE="0.3"))
    assert r.returncode == 75 and "watchdog: phase 'trial step'" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


@pytest.mark.parametrize("where", ["construct", "trial"])
def test_a_rank_whose_managed_exchange_fails_goes_back_to_torch_distributed(where, monkeypatch):
    """bench.py's N > 1 default lets the LIBRARY enqueue RCCL (raw communicators, never run between two devices on this pool): if
    a communicator does not come up, or the first step fails, the run continues with torch.distributed's collectives and the
    redundant or torch-gathered chain -- and the line says which it was"""
    monkeypatch.setenv("RGBDR_BENCH_FAIL_MANAGED", where)
    j = run_bench("--slab", "1/4", "--steps", "4", "--warmup", "2")
    assert j["config"]["collectives"] == "torch.distributed" and j["config"]["halo_transport"] == "rccl"
    assert j["slab"]["integrate_ms"] > 0 and j["ms_per_step"] > 0
    monkeypatch.delenv("RGBDR_BENCH_FAIL_MANAGED")
    j = run_bench("--slab", "1/4", "--steps", "4", "--warmup", "2")
    assert j["config"]["collectives"].startswith("library-managed RCCL")
    # ONE RCCL in the process: the copy torch had mapped already, named with its version; the communicator reports its ranks
    rc = j["config"]["rccl"]
    assert len(rc["copies_mapped"]) == 1 and rc["path"] == rc["copies_mapped"][0] and rc["version"] > 20000 and "RTLD_NOLOAD" in rc["bound"]
    assert j["config"]["rccl_ranks"] == rc["ranks"] == 1          # one GPU stands in for its neighbours: a one-rank communicator


def test_the_chain_schedule_is_chosen_by_measurement():
    """sharded or redundant pre_* chain: the run times both on its own ranks and keeps the faster one; the line says which
    and carries both times (one GPU standing in for rank 1 of 4: the gather goes to the GPU itself, so sharding wins here)"""
    j = run_bench("--slab", "1/4", "--steps", "4", "--warmup", "2")
    ch = j["config"]["pre_chain_choice"]
    assert ch["kept"] in ("sharded", "redundant", "lagged") and ch["steps_each"] >= 8
    times = {k: ch["ms_per_step_" + k] for k in ("sharded", "redundant", "lagged") if ch["ms_per_step_" + k] is not None}
    plain = min(("sharded", "redundant"), key=times.get)
    assert len(times) == 3 and ch["kept"] == ("lagged" if times["lagged"] < 0.98 * times[plain] else plain)
    assert j["config"]["pre_chain"].startswith("sharded by sensor") == (ch["kept"] != "redundant")
    j = run_bench("--slab", "1/4", "--steps", "4", "--warmup", "2", env={"RGBDR_BENCH_CHAIN": "lagged"})
    assert j["config"]["pre_chain_choice"]["kept"] == "lagged" and "one frame ahead of the sweep" in j["config"]["pre_chain"]
    assert j["slab"]["integrate_ms"] > 0 and j["value"] > 0 and "error" not in j["post_pass"] and "error" not in j["bricked"]
    j = run_bench("--slab", "1/4", "--steps", "4", "--warmup", "2", env={"RGBDR_BENCH_CHAIN": "redundant"})
    assert j["config"]["pre_chain_choice"]["kept"] == "redundant" and j["config"]["pre_chain"] == "every sensor on every rank"
    assert j["slab"]["integrate_ms"] > 0 and "error" not in j["post_pass"]


def test_one_slab_of_config_3_as_its_rank_would_run_it():
    j = run_bench("--slab", "0/4", "--steps", "6", "--warmup", "2")
    s = j["slab"]
    assert (s["rank"], s["of"], s["owned_z_rows"], s["faces_staged"]) == (0, 4, 128, 1)
    assert s["integrate_ms"] > 0 and s["integrate_ms_without_staging"] > 0 and 0.4 < s["roofline_frac"] < 1.0
    assert j["config"]["baseline_config"].startswith("configs[3]") and j["config"]["sensors"] == 8
    assert j["roofline"]["bytes_per_launch"] == 512 * 512 * 128 * (4 + 12 * 8) + 8 * 512 * 424 * 8
    assert "true>" in j["roofline"]["kernel"] and j["config"]["halo_transport"] == "rccl"


def test_eight_ranks_run_config_4_end_to_end_on_one_gpu():
    """`python3 bench.py --gpus 8`, the driver's command for BASELINE configs[4] (8 sensors, 1024^3, eight Z slabs with two
    halo tile layers per face, slab ray-march + hole filling; the headline is its fixed-work twin with 4 sensors), with every rank on the one GPU of
    the box and the halos through the host (gloo): the numbers mean nothing, the whole multi-rank code path runs"""
    j = run_bench("--gpus", "8", "--backend", "gloo", "--steps", "2", "--warmup", "1")
    assert j["n_gpus"] == 8 and j["scaling"] == "weak" and j["config"]["grid"] == [1024, 1024, 1024] and j["config"]["sensors"] == 4
    assert j["config"]["baseline_config"].startswith("configs[2] at fixed work per GPU")
    pr = j["per_rank"]
    assert len(pr["integrate_ms"]) == 8 and all(v > 0 for v in pr["integrate_ms"]) and all(v is not None for v in pr["halo_ms"])
    assert j["halo"]["layers_per_face"] == 2 and j["halo"]["bytes_per_face"] == 2 * 128 * 128 * 2048
    assert j["roofline"]["bytes_per_launch"] == 1024 * 1024 * 128 * (4 + 12 * 4) + 4 * 512 * 424 * 8
    assert "error" not in j["post_pass"] and j["post_pass"]["slab_raymarch_composited_ms"] > 0 and j["post_pass"]["surface_pixels"] > 0.05
    b = j["baseline_configs_run"]
    assert "error" not in b and b["baseline_config"].startswith("configs[4]") and b["grid"] == [1024, 1024, 1024] and b["sensors"] == 8
    assert len(b["per_rank"]["integrate_ms"]) == 8
