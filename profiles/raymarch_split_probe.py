#!/usr/bin/env python3
"""k_raymarch<0> (march + shade in one kernel) against k_raymarch<1> (find) + k_raymarch<2> (shade) on the whole benchmark
volume at 1280 x 720: run under `rocprofv3 --kernel-trace --stats` and read the three kernels' averages."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
load_package()
from rgbd_recon_amd import capi, synth
N, W, H, G = 4, 512, 424, 512
scene = synth.Scene(N, W, H, lut_res=(128, 106, 128))
ctx = capi.Context(capi.make_config(N, (W, H), voxel_size=2.0 / G, brick_size=8 * 2.0 / G), 0)
for i in range(N):
    ctx.set_calibration(i, scene.xyz[i], scene.lut_res, scene.uv[i], scene.lut_res, (0.5, 4.5))
    ctx.synth_inverse_calibration(i, scene.pinhole(i))
ctx.step(scene.depth, scene.color)
view = capi.make_view((2.4, 1.8, 2.1), (0.0, 0.7, 0.0), (0.0, 1.0, 0.0), 45.0, 1280, 720, synth.BBOX_MIN, synth.BBOX_MAX)
for skip in (0, 1):
    view.skip_space = skip
    for _ in range(8):
        ctx.raymarch(view)
    for _ in range(8):
        ctx.raymarch_find(view)
        ctx.raymarch_shade(view)
ctx.close()
