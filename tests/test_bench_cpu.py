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
