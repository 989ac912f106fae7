#!/usr/bin/env python3
"""12 full-sweep frames of the headline job (4 x 512 x 424 into 512^3, 1:1 inverse LUT) on the ring scene or, with
RGBDR_PRE_LAYOUT=dense, on the dense scene (every pixel valid and inside the box): the workload of
profiles/pmc_integrate_scenes.sh -- is k_integrate_tiled data dependent, and through what?"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
import torch  # noqa: E402,F401
from rgbd_recon_amd import capi, synth  # noqa: E402

N, W, H, G = 4, 512, 424, 512
os.environ.setdefault("RGBDR_ARENA_TRIALS", "1")     # counters, not times: placement does not matter
scene = synth.Scene(N, W, H, lut_res=(128, 106, 128), seed=1234, layout=os.environ.get("RGBDR_PRE_LAYOUT", "ring"))
ctx = capi.Context(capi.make_config(N, (W, H), voxel_size=2.0 / G, brick_size=8 * 2.0 / G), 0)
for i in range(N):
    ctx.set_calibration(i, scene.xyz[i], scene.lut_res, scene.uv[i], scene.lut_res, (0.5, 4.5))
    ctx.synth_inverse_calibration(i, scene.pinhole(i))
ctx.set_use_bricks(False)
ctx.update(scene.depth, scene.color)
for _ in range(12):
    ctx.clear_occupied_bricks()
    ctx.process_textures()
    ctx.update_occupied_bricks()
    ctx.integrate()
ctx.sync()
ctx.close()
