#!/usr/bin/env python3
"""40 frames of the pre_* chain on the benchmark frame set (4 x 512 x 424): the workload of profiles/pmc_pre.sh.
RGBDR_PRE_LAYOUT=dense: the dense scene instead (every pixel valid and inside the box, synth.Scene(layout="dense"))"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
import torch  # noqa: E402,F401
from rgbd_recon_amd import capi, synth  # noqa: E402

N, W, H, G = 4, 512, 424, 128
scene = synth.Scene(N, W, H, lut_res=(128, 106, 128), seed=1234, layout=os.environ.get("RGBDR_PRE_LAYOUT", "ring"))
ctx = capi.Context(capi.make_config(N, (W, H), voxel_size=2.0 / G, brick_size=8 * 2.0 / G), 0)
for i in range(N):
    ctx.set_calibration(i, scene.xyz[i], scene.lut_res, scene.uv[i], scene.lut_res, (0.5, 4.5))
ctx.update(scene.depth, scene.color)
for _ in range(40):
    ctx.clear_occupied_bricks()
    ctx.process_textures()
    ctx.update_occupied_bricks()
ctx.sync()
ctx.close()
