"""Developer probe: is the integrate launch time a property of the process, of the
context (allocation placement) or of time?  Several contexts per process, several
batches per context, interleaved."""
import os, sys, time
sys.path.insert(0, os.getcwd())
from __graft_entry__ import load_package
load_package()
from rgbd_recon_amd import capi, synth
import torch
N, W, H, G = 4, 512, 424, 512
scene = synth.Scene(N, W, H, lut_res=(128, 106, 128))
def make():
    ctx = capi.Context(capi.make_config(N, (W, H), voxel_size=2.0 / G, brick_size=8 * 2.0 / G), 0)
    for i in range(N):
        ctx.set_calibration(i, scene.xyz[i], scene.lut_res, scene.uv[i], scene.lut_res, (0.5, 4.5))
        ctx.synth_inverse_calibration(i, scene.pinhole(i))
    ctx.step(scene.depth, scene.color)
    ctx.set_use_bricks(False)
    ctx.enable_timer_accumulation(True)
    return ctx
def batch(ctx, n=30):
    ctx.enable_timer_accumulation(True)
    for _ in range(n):
        ctx.integrate()
    ctx.sync()
    ns, k = ctx.timer_stats("2integrate")
    return round(ns / k * 1e-6, 4)
ctxs = []
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    c = make()
    ctxs.append(c)
    print("ctx", k, hex(c.device_tsdf().base), [batch(c) for _ in range(3)], "arena probe", c.arena_probe(), flush=True)
print("revisit", [[batch(c) for _ in range(2)] for c in ctxs], flush=True)
