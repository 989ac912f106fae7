"""Developer probe: cost of filling the halo staging set from the sweep kernel vs a copy after it."""
import os, sys, time
sys.path.insert(0, os.getcwd())
from __graft_entry__ import load_package
load_package()
from rgbd_recon_amd import capi, synth, dist as rdist
import torch
N, W, H = 4, 512, 424
scene = synth.Scene(N, W, H, lut_res=(128, 106, 128))
grid = (1024, 1024, 1024)
ctx = capi.Context(capi.make_config(N, (W, H), voxel_size=2.0 / 1024, brick_size=8 * 2.0 / 1024, res_override=grid, slab_rank=3, slab_count=8), 0)
for i in range(N):
    ctx.set_calibration(i, scene.xyz[i], scene.lut_res, scene.uv[i], scene.lut_res, (0.5, 4.5))
    ctx.synth_inverse_calibration(i, scene.pinhole(i))
dev = torch.device("cuda:0")
d = torch.from_numpy(scene.depth).to(dev); c = torch.from_numpy(scene.color).to(dev)
ctx.set_use_bricks(False)
for b in range(2): ctx.halo_staging(b)
views = rdist.halo_views(ctx.device_tsdf(), dev)
stage = [torch.empty_like(views[0]), torch.empty_like(views[1])]
print("halo layers", ctx.geo.halo_tile_layers, "face MiB", views[0].numel() * 4 / 2**20)
def step(mode):
    ctx.update_device(d.data_ptr(), c.data_ptr()); ctx.clear_occupied_bricks(); ctx.process_textures(); ctx.update_occupied_bricks()
    ctx.integrate()
    if mode == "torch-copy":
        stage[0].copy_(views[0], non_blocking=True); stage[1].copy_(views[1], non_blocking=True)
def run(mode, n=100):
    ctx.set_halo_staging(0 if mode == "kernel" else -1)
    for _ in range(10): step(mode)
    ctx.sync(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step(mode)
    ctx.sync(); torch.cuda.synchronize(); return round((time.perf_counter() - t0) / n * 1e3, 4)
ctx.step(scene.depth, scene.color); ctx.settle(3.0)
for mode in ("none", "torch-copy", "kernel", "none", "torch-copy", "kernel"):
    print(mode, run(mode), flush=True)
