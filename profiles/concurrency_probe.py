#!/usr/bin/env python3
"""How much of the pre_* chain's time is idle machine?  K independent contexts (each with its own HIP stream)
run their chains concurrently; per-frame time of the aggregate vs one context alone."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
load_package()
import torch  # noqa
from rgbd_recon_amd import capi, synth
N, W, H, G = 4, 512, 424, 64
scene = synth.Scene(N, W, H, lut_res=(128, 106, 128), seed=1234)
def make():
    ctx = capi.Context(capi.make_config(N, (W, H), voxel_size=2.0 / G, brick_size=8 * 2.0 / G), 0)
    for i in range(N):
        ctx.set_calibration(i, scene.xyz[i], scene.lut_res, scene.uv[i], scene.lut_res, (0.5, 4.5))
    ctx.update(scene.depth, scene.color)
    return ctx
ctxs = [make() for _ in range(4)]
def run(k, n):
    for _ in range(n):
        for c in ctxs[:k]:
            c.clear_occupied_bricks(); c.process_textures(); c.update_occupied_bricks()
    for c in ctxs[:k]:
        c.sync()
out = {}
for k in (1, 2, 4, 1, 2, 4):
    run(k, 5)
    t0 = time.perf_counter()
    run(k, 200)
    out.setdefault(k, []).append(round((time.perf_counter() - t0) / (200 * k) * 1e3, 4))
print(json.dumps({"ms_per_frame_with_k_concurrent_chains": out}))
