#!/usr/bin/env python3
"""Brick-skipping frame (clear + process_textures + update_occupied + integrate with bricks) at the benchmark
configuration (4 x 512 x 424 -> 512^3, 8-voxel bricks) or the reference's default operating point, for contexts
created under different values of one library knob ("-" = unset): "2integrate" / "1preprocess" by HIP events and
the frame period by wall clock, interleaved rounds, TSDF compared bit for bit between the contexts.
usage: python profiles/brick_probe.py [rounds] [KNOB value value ...]   (RGBDR_PROBE_DEFAULTS=1: 200x221x200 grid)"""
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

N, W, H = 4, 512, 424
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
KNOB = sys.argv[2] if len(sys.argv) > 2 else "RGBDR_NONE"
VALS = tuple(sys.argv[3:]) if len(sys.argv) > 3 else ("-",)
defaults = os.environ.get("RGBDR_PROBE_DEFAULTS") == "1"
scene = synth.Scene(N, W, H, lut_res=(128, 106, 128), seed=1234)
ctxs = {}
for k in VALS:
    os.environ.pop(KNOB, None)
    if k != "-":
        os.environ[KNOB] = k
    if defaults:
        cfg = capi.make_config(N, (W, H), voxel_size=0.01, brick_size=0.1, bbox_min=(-1.0, 0.0, -1.0), bbox_max=(1.0, 2.2, 1.0))
    else:
        cfg = capi.make_config(N, (W, H), voxel_size=2.0 / 512, brick_size=8 * 2.0 / 512)
    c = capi.Context(cfg, 0)
    g = c.geo
    for i in range(N):
        c.set_calibration(i, scene.xyz[i], scene.lut_res, scene.uv[i], scene.lut_res, (0.5, 4.5))
        if defaults:   # 0.007 m inverse LUTs, generated on the device and resampled to the grid at upload
            c.set_inverse_calibration(i, c.generate_inverse_lut(i, (286, 315, 286)), (286, 315, 286))
        else:
            c.synth_inverse_calibration(i, scene.pinhole(i))
    c.set_use_bricks(True)
    c.update(scene.depth, scene.color)
    ctxs[k] = c


def frame(c, n):
    for _ in range(n):
        c.clear_occupied_bricks()
        c.process_textures()
        c.update_occupied_bricks()
        c.integrate()


vols = {}
for k in VALS:
    frame(ctxs[k], 3)
    vols[k] = ctxs[k].readback_tsdf()
same = all(np.array_equal(vols[VALS[0]].view(np.uint32), vols[k].view(np.uint32)) for k in VALS[1:])
occ = len(ctxs[VALS[0]].get_occupied()[0])
res = {k: {"wall": [], "int": [], "pre": []} for k in VALS}
for r in range(rounds):
    for k in VALS:
        c = ctxs[k]
        frame(c, 5)
        c.sync()
        t0 = time.perf_counter()
        frame(c, 200)
        c.sync()
        res[k]["wall"].append((time.perf_counter() - t0) / 200 * 1e3)
        c.enable_timers(True)
        c.set_timer_detail(1)
        c.enable_timer_accumulation(True)
        frame(c, 50)
        c.sync()
        for nm, key in (("2integrate", "int"), ("1preprocess", "pre")):
            ns, n = c.timer_stats(nm)
            res[k][key].append(ns / max(n, 1) * 1e-6)
        c.enable_timer_accumulation(False)
        c.enable_timers(False)
print(json.dumps({"tsdf_bit_identical": bool(same), "occupied_bricks": occ, "grid": list(ctxs[VALS[0]].geo.res_volume),
                  "knob": KNOB, "ms": {k: {m: round(float(np.median(v)), 4) for m, v in d.items()} for k, d in res.items()}}))
