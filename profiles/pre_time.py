#!/usr/bin/env python3
"""Wall-clock time of the pre_* chain alone (4 x 512 x 424 benchmark frames, 400 frames after 40 warm-up) and of the
reference-default / brick-skipping frames; A/B with RGBDR_NO_BOUNDARY_FUSION=1 in the environment."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
import torch  # noqa: E402,F401
from rgbd_recon_amd import capi, synth  # noqa: E402

if os.environ.get("RGBDR_PROBE_LIB"):       # A/B against a developer build (e.g. profiles/probes_src/librgbdr_hip_nolab.so)
    capi.LIB_PATH = os.path.join(ROOT, os.environ["RGBDR_PROBE_LIB"])

N, W, H, G = 4, 512, 424, 128
scene = synth.Scene(N, W, H, lut_res=(128, 106, 128), seed=1234)
ctx = capi.Context(capi.make_config(N, (W, H), voxel_size=2.0 / G, brick_size=8 * 2.0 / G), 0)
for i in range(N):
    ctx.set_calibration(i, scene.xyz[i], scene.lut_res, scene.uv[i], scene.lut_res, (0.5, 4.5))
ctx.update(scene.depth, scene.color)


def chain():
    ctx.clear_occupied_bricks()
    ctx.process_textures()
    ctx.update_occupied_bricks()


for _ in range(40):
    chain()
ctx.sync()
best = 1e9
for rep in range(5):
    t0 = time.perf_counter()
    for _ in range(400):
        chain()
    ctx.sync()
    best = min(best, (time.perf_counter() - t0) / 400 * 1e3)
print("pre_* chain %.4f ms per frame (best of 5 x 400), boundary fusion %s" % (best, "off" if os.environ.get("RGBDR_NO_BOUNDARY_FUSION") else "on"))
ctx.close()
