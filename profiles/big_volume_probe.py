#!/usr/bin/env python3
"""One sensor into 2048 x 2048 x 1024 voxels (2^32: 17 GB of TSDF, 52 GB of LUT planes) on one MI355X: full-sweep time and rate.
    python3 profiles/big_volume_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RGBDR_ARENA_TRIALS", "1")
from __graft_entry__ import load_package
load_package()
import torch  # noqa
from rgbd_recon_amd import capi, synth
N, W, H = int(sys.argv[1]) if len(sys.argv) > 1 else 1, 512, 424
grid = (2048, 2048, 1024)
scene = synth.Scene(N, W, H, lut_res=(128, 106, 128), seed=1234)
ctx = capi.Context(capi.make_config(N, (W, H), voxel_size=2.0 / grid[0], brick_size=8 * 2.0 / grid[0], res_override=grid), 0)
for i in range(N):
    ctx.set_calibration(i, scene.xyz[i], scene.lut_res, scene.uv[i], scene.lut_res, (0.5, 4.5))
    ctx.synth_inverse_calibration(i, scene.pinhole(i))
ctx.set_use_bricks(False)
ctx.enable_timer_accumulation(True)
for _ in range(3):
    ctx.step(scene.depth, scene.color)
ctx.sync(); ctx.timer_stats("2integrate")
t0 = time.perf_counter()
steps = 20
for _ in range(steps):
    ctx.step(scene.depth, scene.color)
ctx.sync()
dt = (time.perf_counter() - t0) / steps
ns, cnt = ctx.timer_stats("2integrate")
V = grid[0] * grid[1] * grid[2]
b = V * (4 + 12 * N) + N * W * H * 8
print({"sensors": N, "grid": grid, "ms_per_frame": round(dt * 1e3, 3), "integrate_ms": round(ns / cnt * 1e-6, 3), "Gvoxels_per_s": round(V / dt / 1e9, 1),
       "algorithmic_GB": round(b / 1e9, 2), "GBps": round(b / (ns / cnt * 1e-9) / 1e9, 1), "frac_of_8TBps": round(b / (ns / cnt * 1e-9) / 8e12, 4)})
ctx.close()
