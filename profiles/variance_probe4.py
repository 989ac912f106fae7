#!/usr/bin/env python3
"""Developer probe: why does bench_legs.measure_modes read an integrate launch 2-3 % longer than the headline on the same
context?  Times the full sweep (HIP events around the launch, 20 steps) in a sequence of situations."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
load_package()
import torch
from rgbd_recon_amd import capi, synth
N, W, H, G = 4, 512, 424, 512
scene = synth.Scene(N, W, H, lut_res=(128, 106, 128), seed=1234)
dev = torch.device("cuda", 0)
A = (torch.from_numpy(scene.depth).to(dev), torch.from_numpy(scene.color).to(dev))
ctx = capi.Context(capi.make_config(N, (W, H), voxel_size=2.0 / G, brick_size=8 * 2.0 / G, flags=capi.FLAGS_DEFAULT), 0)
for i in range(N):
    ctx.set_calibration(i, scene.xyz[i], scene.lut_res, scene.uv[i], scene.lut_res, (0.5, 4.5))
    ctx.synth_inverse_calibration(i, scene.pinhole(i))
B = (torch.from_numpy(scene.depth).to(dev), torch.from_numpy(scene.color).to(dev))
torch.cuda.synchronize()
ctx.settle(3.0)


def run(tag, frame, steps=20, warm=5, pre=None):
    ctx.set_use_bricks(False)
    if pre:
        pre()
    def step():
        ctx.update_device(frame[0].data_ptr(), frame[1].data_ptr())
        ctx.clear_occupied_bricks(); ctx.process_textures(); ctx.update_occupied_bricks(); ctx.integrate()
    for _ in range(warm):
        step()
    ctx.sync()
    ctx.set_timer_detail(0); ctx.enable_timer_accumulation(True)
    for _ in range(steps):
        step()
    ctx.sync()
    ns, n = ctx.timer_stats("2integrate")
    ctx.enable_timer_accumulation(False); ctx.enable_timers(False); ctx.set_timer_detail(2)
    print("%-58s integrate %.4f ms" % (tag, ns / max(n, 1) * 1e-6))


run("frame allocated before the context", A)
run("frame allocated after the context", B)
run("again: before", A)
def skip_on_off():
    ctx.set_skip_background(True)
    for _ in range(3):
        ctx.update_device(A[0].data_ptr(), A[1].data_ptr()); ctx.clear_occupied_bricks(); ctx.process_textures(); ctx.update_occupied_bricks(); ctx.integrate()
    ctx.sync(); ctx.set_skip_background(False)
run("after a background-skip episode", A, pre=skip_on_off)
def bricks_on_off():
    ctx.set_use_bricks(True)
    for _ in range(3):
        ctx.update_device(A[0].data_ptr(), A[1].data_ptr()); ctx.clear_occupied_bricks(); ctx.process_textures(); ctx.update_occupied_bricks(); ctx.integrate()
    ctx.sync(); ctx.set_use_bricks(False)
run("after a brick-sweep episode", A, pre=bricks_on_off)
import time
def burst():
    for i in range(1000):
        ctx.update_device(A[0].data_ptr(), A[1].data_ptr()); ctx.clear_occupied_bricks(); ctx.process_textures(); ctx.update_occupied_bricks(); ctx.integrate()
        if i % 64 == 63: ctx.sync()
    ctx.sync()
run("right after a 1.1 s burst", A, pre=burst)
time.sleep(2.0)
run("2 s of idle later", A)
c2 = capi.Context(capi.make_config(1, (W, H), voxel_size=2.0 / 128, brick_size=8 * 2.0 / 128), 0)
run("with a second (small) context alive", A)
c2.close()
run("after closing it", A)
ctx.close()
