#!/usr/bin/env python3
"""In-process timing of the pre_* chain (clear + process_textures + update_occupied per frame on the benchmark
frame set) for contexts created under different values of one environment knob the library reads at context
creation (RGBDR_SEPARATE_PASSES ...; "-" = unset), interleaved rounds, with a bit-for-bit
comparison of every image / the brick counters between the contexts.
usage: python profiles/pre_probe.py [rounds] [KNOB value value ...]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
import torch  # noqa: E402,F401
from rgbd_recon_amd import capi, synth  # noqa: E402

N, W, H, G = 4, 512, 424, 64
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
scene = synth.Scene(N, W, H, lut_res=(128, 106, 128), seed=1234)
KNOB = sys.argv[2] if len(sys.argv) > 2 else "RGBDR_SEPARATE_PASSES"
LANES = tuple(sys.argv[3:]) if len(sys.argv) > 3 else ("-", "1")
ctxs = {}
for k in LANES:
    os.environ.pop(KNOB, None)
    if k != "-":
        os.environ[KNOB] = k
    c = capi.Context(capi.make_config(N, (W, H), voxel_size=2.0 / G, brick_size=8 * 2.0 / G), 0)
    for i in range(N):
        c.set_calibration(i, scene.xyz[i], scene.lut_res, scene.uv[i], scene.lut_res, (0.5, 4.5))
    c.update(scene.depth, scene.color)
    ctxs[k] = c


def chain(c, n):
    for _ in range(n):
        c.clear_occupied_bricks()
        c.process_textures()
        c.update_occupied_bricks()


def images(c):
    out = {w: [c.readback_image(w, i) for i in range(N)] for w in range(1, 8)}
    out["counters"] = c.readback_brick_counters()
    return out


imgs = {}
for k in LANES:
    chain(ctxs[k], 3)          # several frames: the counters must be cleared and rebuilt identically each time
    imgs[k] = images(ctxs[k])
same = all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for k in LANES[1:] for w in range(1, 8)
           for a, b in zip(imgs[LANES[0]][w], imgs[k][w]))
same = same and all(np.array_equal(imgs[LANES[0]]["counters"], imgs[k]["counters"]) for k in LANES[1:])
wall = {k: [] for k in LANES}
for r in range(rounds):
    for k in LANES:
        chain(ctxs[k], 5)
        ctxs[k].sync()
        t0 = time.perf_counter()
        chain(ctxs[k], 200)
        ctxs[k].sync()
        wall[k].append((time.perf_counter() - t0) / 200 * 1e3)
print(json.dumps({"images_and_counters_bit_identical": bool(same), "counters_sum": int(imgs[LANES[0]]["counters"].sum()),
                  "knob": KNOB, "chain_ms_per_frame": {k: {"median": round(float(np.median(w)), 4), "min": round(min(w), 4)}
                                                  for k, w in wall.items()}}))
