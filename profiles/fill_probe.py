#!/usr/bin/env python3
"""Hole filling of a 1280 x 720 ray-marched frame of the benchmark volume: median of the "holefill" timer."""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
from rgbd_recon_amd import capi, synth  # noqa: E402

N, W, H, G = 4, 512, 424, 512
scene = synth.Scene(N, W, H, lut_res=(128, 106, 128))
ctx = capi.Context(capi.make_config(N, (W, H), voxel_size=2.0 / G, brick_size=8 * 2.0 / G), 0)
for i in range(N):
    ctx.set_calibration(i, scene.xyz[i], scene.lut_res, scene.uv[i], scene.lut_res, (0.5, 4.5))
    ctx.synth_inverse_calibration(i, scene.pinhole(i))
ctx.step(scene.depth, scene.color)
ctx.set_timer_detail(2)
ctx.enable_timers(True)
view = capi.make_view((2.4, 1.8, 2.1), (0.0, 0.7, 0.0), (0.0, 1.0, 0.0), 45.0, 1280, 720, synth.BBOX_MIN, synth.BBOX_MAX)
view.skip_space = 1
ctx.raymarch(view)
t = []
for _ in range(25):
    ctx.fill_colors(1280, 720)
    t.append(ctx.timer_ns("holefill") * 1e-6)
print("holefill %.4f ms (median of 25; min %.4f)" % (statistics.median(t), min(t)))
# back to back without the readback (the frame stays on the device, as in a display loop): clocks stay up
import time  # noqa: E402
L = capi.lib()
for _ in range(3000):
    L.rgbdr_fill_colors(ctx._h, None, None)
t = []
t0 = time.perf_counter()
for _ in range(500):
    L.rgbdr_fill_colors(ctx._h, None, None)
    t.append(ctx.timer_ns("holefill") * 1e-6)
wall = (time.perf_counter() - t0) / 500 * 1e3
print("holefill back to back %.4f ms (median of 500 event timings; min %.4f; wall clock per call incl. the sync %.4f)"
      % (statistics.median(t), min(t), wall))
ctx.close()
