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
    """the driver's command line; in this container there is no HIP device, so the ranks refuse and the parent
    reports it with a non-zero status instead of hanging or printing a line"""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("CPU-only check")
    r = subprocess.run([sys.executable, "-X", "importtime", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "needs a HIP device" in r.stderr and "[bench launcher] rank" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    # -X importtime lists what the PARENT imported (the ranks are started without it): no torch, so no GPU runtime
    imported = [l.split("|")[-1].strip() for l in r.stderr.splitlines() if l.startswith("import time:")]
    assert "json" in imported and "torch" not in imported


def test_slab_argument():
    b = load_bench()
    assert b.parse_slab("1/4") == (1, 4) and b.parse_slab("7/8") == (7, 8)
    import pytest
    for bad in ("4/4", "1", "0/3", "a/b"):
        with pytest.raises(SystemExit):
            b.parse_slab(bad)
