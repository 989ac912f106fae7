#!/usr/bin/env python3
"""Developer probe: the slab timelines show the sweep 8 per cent shorter when RCCL's gather kernel runs beside it on another
queue (r05_lag_timeline_*.txt).  Is that the company?  Times the headline sweep alone and with a spin kernel of 1, 8, 64 and
256 workgroups on a side stream."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
load_package()
import torch
from rgbd_recon_amd import capi, synth
N = int(os.environ.get("SENSORS", "4"))
W, H, G = 512, 424, 512
scene = synth.Scene(N, W, H, lut_res=(128, 106, 128), seed=1234)
dev = torch.device("cuda", 0)
ctx = capi.Context(capi.make_config(N, (W, H), voxel_size=2.0 / G, brick_size=8 * 2.0 / G, flags=capi.FLAGS_DEFAULT), 0)
for i in range(N):
    ctx.set_calibration(i, scene.xyz[i], scene.lut_res, scene.uv[i], scene.lut_res, (0.5, 4.5))
    ctx.synth_inverse_calibration(i, scene.pinhole(i))
A = (torch.from_numpy(scene.depth).to(dev), torch.from_numpy(scene.color).to(dev))
torch.cuda.synchronize()
ctx.settle(3.0)
ctx.set_use_bricks(False)
side = torch.cuda.Stream(device=dev)
spin = torch.zeros(1 << 24, device=dev)


def step(company):
    ctx.update_device(A[0].data_ptr(), A[1].data_ptr())
    ctx.clear_occupied_bricks(); ctx.process_textures(); ctx.update_occupied_bricks()
    if company:
        ev = torch.cuda.Event(); ev.record(torch.cuda.ExternalStream(ctx.stream(), device=dev))
        with torch.cuda.stream(side):
            side.wait_event(ev)
            company()
    ctx.integrate()


def run(tag, company=None, steps=30, warm=8):
    for _ in range(warm):
        step(company)
    ctx.sync(); torch.cuda.synchronize()
    ctx.set_timer_detail(0); ctx.enable_timer_accumulation(True)
    for _ in range(steps):
        step(company)
    ctx.sync(); torch.cuda.synchronize()
    ns, n = ctx.timer_stats("2integrate")
    ctx.enable_timer_accumulation(False); ctx.enable_timers(False); ctx.set_timer_detail(2)
    print("%-58s integrate %.4f ms" % (tag, ns / max(n, 1) * 1e-6), flush=True)


run("alone")
run("beside torch.cuda._sleep (one workgroup, ~0.4 ms)", lambda: torch.cuda._sleep(800000))
if os.environ.get("BRICKS") == "1":      # the brick sweep: 2560 persistent blocks instead of one block per tile
    ctx.set_use_bricks(True)
    run("brick sweep alone")
    run("brick sweep beside torch.cuda._sleep", lambda: torch.cuda._sleep(800000))
    ctx.set_use_bricks(False)
    ctx.set_skip_background(True)        # listed tiles: one block per listed tile
    run("background-skip sweep alone")
    run("background-skip sweep beside torch.cuda._sleep", lambda: torch.cuda._sleep(800000))
    ctx.set_skip_background(False)
for k in (1 << 10, 1 << 14, 1 << 18, 1 << 22):
    v = spin[:k]
    def many(v=v):
        for _ in range(8):
            v.mul_(1.0001)
    run("beside 8 small elementwise kernels over %d floats" % k, many)
run("alone again")
ctx.close()
