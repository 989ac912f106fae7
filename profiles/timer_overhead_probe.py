"""Developer probe: what do the per-pass HIP event records cost per frame?"""
import os, sys, time
sys.path.insert(0, os.getcwd())
from __graft_entry__ import load_package
load_package()
from rgbd_recon_amd import capi, synth
import torch
N, W, H, G = 4, 512, 424, 512
scene = synth.Scene(N, W, H, lut_res=(128, 106, 128))
ctx = capi.Context(capi.make_config(N, (W, H), voxel_size=2.0 / G, brick_size=8 * 2.0 / G), 0)
for i in range(N):
    ctx.set_calibration(i, scene.xyz[i], scene.lut_res, scene.uv[i], scene.lut_res, (0.5, 4.5))
    ctx.synth_inverse_calibration(i, scene.pinhole(i))
d = torch.from_numpy(scene.depth).cuda(); c = torch.from_numpy(scene.color).cuda()
torch.cuda.synchronize()
def step():
    ctx.update_device(d.data_ptr(), c.data_ptr()); ctx.clear_occupied_bricks(); ctx.process_textures(); ctx.update_occupied_bricks(); ctx.integrate()
def run(n=200):
    for _ in range(10): step()
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(n): step()
    ctx.sync(); return (time.perf_counter() - t0) / n * 1e3
ctx.step(scene.depth, scene.color); ctx.settle(3.0)
for bricks in (False, True):
    ctx.set_use_bricks(bricks)
    for mode in ("off", "detail1", "detail1+accumulate", "detail2+accumulate", "off"):
        ctx.enable_timers(mode != "off")
        ctx.set_timer_detail(2 if "detail2" in mode else 1)
        ctx.enable_timer_accumulation("accumulate" in mode)
        print("bricks", bricks, mode, round(run(), 4), flush=True)
        ctx.enable_timer_accumulation(False)
