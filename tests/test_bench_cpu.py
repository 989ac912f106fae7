"""bench.py picks the BASELINE.json config its --gpus value names (no GPU needed for that part)."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_workload_follows_the_baseline_configs():
    b = load_bench()
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))["configs"]
    n, grid, cfg, scaling = b.choose_workload(1)
    assert (n, grid, scaling) == (4, (512, 512, 512), "weak") and cfg.startswith("configs[2]") and "4 sensors, 512" in base[2]
    for world in (2, 4):
        n, grid, cfg, scaling = b.choose_workload(world)
        assert (n, grid, scaling) == (8, (512, 512, 512), "strong") and cfg.startswith("configs[3]")
    assert "8 sensors, 512" in base[3] and "4 MI355X" in base[3]
    n, grid, cfg, scaling = b.choose_workload(8)
    assert (n, grid, scaling) == (8, (1024, 1024, 1024), "weak") and cfg.startswith("configs[4]")
    assert "8 sensors, 1024" in base[4] and "8 MI355X" in base[4]
    # the weak-scaling grids stay available behind --weak, 134 M voxels per GPU
    for world, g in ((2, (512, 512, 1024)), (4, (512, 1024, 1024)), (8, (1024, 1024, 1024))):
        n, grid, cfg, scaling = b.choose_workload(world, weak=True)
        assert n == 4 and grid == g and scaling == "weak" and grid[0] * grid[1] * grid[2] == world * 512 ** 3
    assert b.choose_workload(1, sensors=2, cubic_grid=256)[:2] == (2, (256, 256, 256))


def test_slab_geometry_of_the_multi_gpu_configs(pkg):
    """configs[3] / configs[4]: the slabs are whole tile layers, cover the volume and keep the halo their consumers need"""
    capi = pkg.capi
    b = load_bench()
    for world in (2, 4, 8):
        n, grid, _, _ = b.choose_workload(world)
        covered = 0
        for rank in range(world):
            cfg = capi.make_config(n, (512, 424), voxel_size=2.0 / grid[0], brick_size=8 * 2.0 / grid[0], res_override=grid,
                                   slab_rank=rank, slab_count=world)
            g = capi.compute_geometry(cfg)
            assert g.slab_voxel_z0 == covered and g.halo_tile_layers == (1 if grid[2] == 512 else 2)
            covered = g.slab_voxel_z1
            assert (g.slab_voxel_z1 - g.slab_voxel_z0) * world == grid[2]
        assert covered == grid[2]


# ---- `python3 bench.py --gpus N` started as a plain process (how the driver starts it) spawns its own ranks ----
def test_launcher_gives_every_rank_its_environment(tmp_path):
    import sys
    b = load_bench()
    child = [sys.executable, "-c",
             "import os, sys; open(os.path.join(%r, os.environ['RANK']), 'w').write(' '.join(os.environ[k] for k in "
             "('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'HSA_ENABLE_IPC_MODE_LEGACY'))); "
             "assert 'torch' not in sys.modules" % str(tmp_path)]
    assert b.launch_ranks(3, [], child_cmd=child, timeout=60) == 0
    seen = [open(os.path.join(str(tmp_path), str(r))).read().split() for r in range(3)]
    for r, s in enumerate(seen):
        assert s[:4] == [str(r), str(r), "3", "127.0.0.1"] and s[5] == "0"
    assert len({s[4] for s in seen}) == 1 and int(seen[0][4]) > 0          # one rendezvous port for all


def test_launcher_stops_the_other_ranks_when_one_fails():
    import sys
    import time
    b = load_bench()
    child = [sys.executable, "-c", "import os, sys, time\nif os.environ['RANK'] == '1': sys.exit(3)\ntime.sleep(120)"]
    t0 = time.monotonic()
    assert b.launch_ranks(3, [], child_cmd=child, timeout=100) == 3
    assert time.monotonic() - t0 < 30
    sleeper = [sys.executable, "-c", "import time; time.sleep(120)"]
    t0 = time.monotonic()
    assert b.launch_ranks(2, [], child_cmd=sleeper, timeout=1.0) == 124 and time.monotonic() - t0 < 30


def test_plain_multi_gpu_command_fails_loudly_without_a_gpu_and_never_touches_torch_in_the_parent():
    """the driver's command line; in this container there is no HIP device, so the ranks of every rung refuse and rank 0's
    supervisor reports it: ONE line with "error" and the attempts, a non-zero status -- no hang, no silent end"""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("CPU-only check")
    r = subprocess.run([sys.executable, "-X", "importtime", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "needs a HIP device" in r.stderr and "[bench launcher] rank" in r.stderr
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and "every rung" in lines[0]["error"] and "value" not in lines[0]
    assert [a["rung"] for a in lines[0]["attempts"]] == [0, 1, 2, 3] and lines[0]["attempts"][2]["flags"] == b_rung_flags(2)
    assert lines[0]["attempts"][3]["flags"][:2] == ["--halo-transport", "peer"]      # the last rung needs no RCCL per frame
    # -X importtime lists what the PARENT imported (supervisors and ranks are started without it): no torch, so no GPU runtime
    imported = [l.split("|")[-1].strip() for l in r.stderr.splitlines() if l.startswith("import time:")]
    assert "json" in imported and "torch" not in imported


def b_rung_flags(k):
    return load_bench().RUNGS[k][1]


# ---- the launch ladder (VERDICT r4 task 1): an N > 1 run cannot end without a JSON line ---------------------------------
STUB = [__import__("sys").executable, os.path.join(ROOT, "tests", "ladder_stub.py")]


def run_ladder(plan, tmp_path, *extra, via_torchrun=False, budgets="5,5,5", timeout=120, n=2):
    import subprocess
    import sys
    import time
    env = dict(os.environ, RGBDR_BENCH_CHILD_CMD=json.dumps(STUB), STUB_PLAN=json.dumps(plan), STUB_LOG=str(tmp_path))
    env.pop("WORLD_SIZE", None)
    args = [os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "3", "--rung-budgets", budgets] + list(extra)
    if via_torchrun:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
               "--master-port", str(load_bench().free_port())] + args
    else:
        cmd = [sys.executable] + args
    t0 = time.monotonic()
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    took = time.monotonic() - t0
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (r.stdout[-2000:], r.stderr[-3000:])
    return r.returncode, lines[0], took, r.stderr


def test_ladder_first_rung_hangs_second_succeeds(tmp_path):
    """rung 0 (library-managed RCCL) hangs on every rank: its budget ends it, FRESH children run rung 1 with
    --torch-collectives on a new rendezvous port, and the one line says which rung produced it and what failed before"""
    rc, line, took, err = run_ladder(["hang", "ok", "ok"], tmp_path)
    assert rc == 0 and line["value"] == 101.0 and line["legs"] == "done" and "provisional" not in line
    la = line["launch"]
    assert la["rung"] == 1 and la["rung_flags"] == ["--torch-collectives", "--no-lagged"] and la["line"] == "final" and la["launched_by"] == "bench.py"
    assert len(la["failed_attempts"]) == 1 and la["failed_attempts"][0]["rung"] == 0 and "budget" in la["failed_attempts"][0]["outcome"]
    assert took < 40
    seen = {f: json.load(open(os.path.join(str(tmp_path), f))) for f in os.listdir(str(tmp_path))}
    assert set(seen) == {"rung0.rank0", "rung0.rank1", "rung1.rank0", "rung1.rank1"}
    assert seen["rung1.rank0"]["argv"][-2:] == ["--torch-collectives", "--no-lagged"] and "--torch-collectives" not in seen["rung0.rank0"]["argv"]
    assert seen["rung0.rank0"]["port"] == seen["rung0.rank1"]["port"] != seen["rung1.rank0"]["port"] == seen["rung1.rank1"]["port"]


def test_ladder_under_torch_distributed_run(tmp_path):
    """the same when torch.distributed.run starts the ranks (how the driver starts an N > 1 run): each rank it starts is a
    supervisor; the children rendezvous on their own port, not through the agent's store"""
    rc, line, took, err = run_ladder(["fail", "hang", "ok"], tmp_path, via_torchrun=True, timeout=180)
    assert rc == 0 and line["value"] == 102.0
    la = line["launch"]
    assert la["rung"] == 2 and la["rung_flags"] == ["--torch-collectives", "--no-shard", "--no-lagged", "--weak"] and la["launched_by"] == "torch.distributed.run"
    assert [a["rung"] for a in la["failed_attempts"]] == [0, 1]
    assert la["failed_attempts"][0]["child_errors"] == ["stub failure on rung 0"] and la["failed_attempts"][0]["child_status"] == 5


def test_ladder_every_rung_fails_ends_with_an_error_line(tmp_path):
    rc, line, took, err = run_ladder(["fail", "fail", "hang", "fail"], tmp_path, budgets="4,4,4,4")
    assert rc != 0 and "error" in line and "value" not in line and [a["rung"] for a in line["attempts"]] == [0, 1, 2, 3]
    assert took < 40


def test_ladder_keeps_the_provisional_line_when_the_legs_hang(tmp_path):
    """the headline exists (rank 0 printed its provisional line) and then a leg hangs: the rung's budget ends the children and
    the line is the provisional one, marked as such -- not a second attempt, not an error"""
    rc, line, took, err = run_ladder(["provisional", "ok", "ok"], tmp_path)
    assert rc == 0 and line["value"] == 100.0 and "legs" not in line and "provisional" not in line
    assert line["launch"]["rung"] == 0 and line["launch"]["line"].startswith("provisional") and line["launch"]["failed_attempts"] == []


def test_ladder_a_dead_rank_ends_the_rung_early(tmp_path):
    """rank 1's child dies at once while rank 0's would wait for ever: the rung is given up after a short grace, long before its
    budget, and the next rung runs"""
    rc, line, took, err = run_ladder(["rank1 dies", "ok", "ok"], tmp_path, budgets="300,20,20")
    assert rc == 0 and line["launch"]["rung"] == 1 and "rank 1" in line["launch"]["failed_attempts"][0]["outcome"]
    assert took < 60


def test_ladder_never_exceeds_the_launch_timeout(tmp_path):
    """--launch-timeout bounds the whole ladder: with 26 s and rungs of 20 s the second rung gets what is left or is not started"""
    rc, line, took, err = run_ladder(["hang", "hang", "hang", "hang"], tmp_path, "--launch-timeout", "26", budgets="20,20,20,20")
    assert rc != 0 and "error" in line and took < 40
    assert "not started" in line["attempts"][-1]["outcome"] or line["attempts"][-1]["seconds"] < 10


def test_slab_argument():
    b = load_bench()
    assert b.parse_slab("1/4") == (1, 4) and b.parse_slab("7/8") == (7, 8)
    import pytest
    for bad in ("4/4", "1", "0/3", "a/b"):
        with pytest.raises(SystemExit):
            b.parse_slab(bad)


def test_file_store_and_watchdog_units(tmp_path, monkeypatch):
    """the two small mechanisms under the ladder: values put by one process are read by another (rename, no partial reads),
    and a phase that overruns ends the PROCESS from the side thread with the configured status"""
    import subprocess
    import sys
    b = load_bench()
    monkeypatch.setenv("RGBDR_BENCH_JOB", "unit_%d" % os.getpid())
    st = b.FileStore()
    assert st.get("k") is None and st.wait("k", timeout=0.2) is None
    st.put("k", {"a": [1, 2, 3]})
    assert b.FileStore().get("k") == {"a": [1, 2, 3]}
    st.cleanup()
    assert not os.path.exists(st.dir)
    code = ("import sys, time; sys.path.insert(0, %r); import importlib.util as u; "
            "sp = u.spec_from_file_location('bm', %r); m = u.module_from_spec(sp); sp.loader.exec_module(m)\n"
            "wd = m.Watchdog(lambda name, budget: sys.stderr.write('expired %%s\\n' %% name))\n"
            "with wd.phase('quick', 5.0): pass\n"
            "with wd.phase('slow', 0.5): time.sleep(30)\n") % (ROOT, os.path.join(ROOT, "bench.py"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert r.returncode == b.EXIT_WATCHDOG and "watchdog: phase 'slow'" in r.stderr and "expired slow" in r.stderr


def test_leg_isolation_units():
    """bench_legs.run_leg without a GPU: a leg that throws leaves {"error": ...} under its key and the rig's switches are
    put back; the legs after it still run; the test hook names a leg to fail"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_legs_mod", os.path.join(ROOT, "bench_legs.py"))
    legs = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(legs)
    b = load_bench()

    class FakeRig:
        rank = 0

        def __init__(self):
            self.restored = 0
            self.watchdog = b.Watchdog(lambda name, budget: None)

        def restore_defaults(self):
            self.restored += 1

    rig, out = FakeRig(), {"roofline": {}}
    legs.run_leg(rig, out, "good", lambda: {"ms": 1.0})
    legs.run_leg(rig, out, "bad", lambda: 1 / 0)
    legs.run_leg(rig, out, "box", lambda: {"w": 3}, into=out["roofline"])
    assert out["good"] == {"ms": 1.0} and out["bad"]["error"].startswith("ZeroDivisionError") and out["roofline"]["box"] == {"w": 3}
    assert rig.restored == 1
    os.environ["RGBDR_BENCH_FAIL_LEG"] = "later"
    try:
        legs.run_leg(rig, out, "later", lambda: {"never": True})
    finally:
        del os.environ["RGBDR_BENCH_FAIL_LEG"]
    assert out["later"] == {"error": "RuntimeError: RGBDR_BENCH_FAIL_LEG=later"} and rig.restored == 2


def test_a_lagged_headline_far_off_its_trial_is_timed_again_on_the_plain_schedule():
    """bench_chain.recheck_lagged_headline without a GPU: the K timed steps of the lagged schedule are kept when they agree
    with its trial (or still beat the plain schedules), otherwise the lagged chain is left and the K steps are timed again"""
    import types
    b = load_bench()

    class FakeLag:
        closed = 0

        def close(self):
            FakeLag.closed += 1

    class FakeCtx:
        def set_sensor_shard(self, first, count):
            self.shard = (first, count)

    def rig_with(choice):
        r = types.SimpleNamespace(args=types.SimpleNamespace(steps=40, warmup=10), chain_choice=dict(choice), lag=FakeLag(),
                                  lag_keep=(None, None), plain_chain=("sharded", "the gather", 2, 2), ctx=FakeCtx(), gather=None, timings=[])
        r.barrier = lambda: None
        r.timed = lambda bricks, steps, warmup: (r.timings.append((steps, warmup)), (steps * 0.65e-3, {"2integrate": (1, 1)}))[1]
        return r

    choice = {"ms_per_step_sharded": 0.70, "ms_per_step_redundant": 0.75, "ms_per_step_lagged": 0.67, "kept": "lagged"}
    r = rig_with(choice)
    dt, stats = b.recheck_lagged_headline(r, 40 * 0.66e-3, "first")          # as on the trial: kept
    assert (dt, stats) == (40 * 0.66e-3, "first") and r.lag is not None and r.timings == []
    r = rig_with(choice)
    dt, stats = b.recheck_lagged_headline(r, 40 * 0.69e-3, "first")          # 3 % off the trial, below the plain ones: kept
    assert stats == "first" and r.chain_choice["kept"] == "lagged"
    r = rig_with(choice)
    dt, stats = b.recheck_lagged_headline(r, 40 * 1.54e-3, "first")          # a stall: discarded
    assert r.lag is None and FakeLag.closed == 1 and r.gather == "the gather" and r.ctx.shard == (2, 2)
    assert r.timings == [(40, 10)] and abs(dt - 40 * 0.65e-3) < 1e-12 and stats == {"2integrate": (1, 1)}
    assert r.chain_choice["kept"] == "sharded" and r.chain_choice["lagged_headline_discarded_ms_per_step"] == 1.54


def test_a_headline_with_a_host_stall_is_timed_again():
    """bench_chain.retime_after_a_host_stall without a GPU: a step that held the host for more than 30 ms -> the K steps are
    timed again (at most twice) and the line keeps what was discarded; a clean run is left alone"""
    import types
    b = load_bench()

    def rig_with(longest_first, then):
        r = types.SimpleNamespace(args=types.SimpleNamespace(steps=40, warmup=10), world=1, longest_host_step_ms=longest_first, timings=[])
        seq = list(then)

        def timed(bricks, steps, warmup):
            r.timings.append((steps, warmup))
            r.longest_host_step_ms = seq.pop(0)
            return steps * 0.65e-3, {"2integrate": (1, 1)}
        r.timed = timed
        return r

    r = rig_with(0.4, [])
    assert b.retime_after_a_host_stall(r, 40 * 0.66e-3, "first") == (40 * 0.66e-3, "first") and not hasattr(r, "retimed")
    r = rig_with(81.8, [0.5])
    dt, stats = b.retime_after_a_host_stall(r, 40 * 1.6e-3, "first")
    assert r.timings == [(40, 10)] and abs(dt - 40 * 0.65e-3) < 1e-12 and stats == {"2integrate": (1, 1)}
    assert r.retimed["discarded"] == [{"ms_per_step": 1.6, "longest_host_step_ms": 81.8}] and r.retimed["longest_host_step_ms_kept"] == 0.5
    r = rig_with(81.8, [55.0, 60.0])                     # never clean: two more attempts, the last one stands
    b.retime_after_a_host_stall(r, 40 * 1.6e-3, "first")
    assert len(r.timings) == 2 and len(r.retimed["discarded"]) == 2
