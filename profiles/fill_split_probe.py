#!/usr/bin/env python3
"""Hole filling of 1280 x 720 frames that are all surface / all hole / half and half: which part of rgbdr_fill_colors is the
copy, which the hole path (run under rocprofv3 --kernel-trace for the per-kernel split, or read the holefill timer)."""
import os
import statistics
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
from rgbd_recon_amd import capi  # noqa: E402

W, H = 1280, 720
ctx = capi.Context(capi.make_config(1, (64, 53), voxel_size=2.0 / 32, brick_size=0.5), 0)
ctx.set_timer_detail(2)
ctx.enable_timers(True)
rng = np.random.default_rng(1)
L = capi.lib()
for name, hole in (("all surface", np.zeros((H, W), bool)), ("all hole", np.ones((H, W), bool)),
                   ("left half hole", np.tile(np.arange(W) < W // 2, (H, 1))),
                   ("disc of surface (r = 200)", (np.add.outer((np.arange(H) - H / 2) ** 2, (np.arange(W) - W / 2) ** 2) > 200 ** 2))):
    col = rng.random((H, W, 4), dtype=np.float32)
    dep = rng.random((H, W), dtype=np.float32)
    col[hole] = np.float32([0, 1, 0, 0])
    dep[hole] = 1.0
    ctx.upload_view_frame(col, dep)
    for _ in range(300):
        L.rgbdr_fill_colors(ctx._h, None, None)
    t = []
    for _ in range(200):
        L.rgbdr_fill_colors(ctx._h, None, None)
        t.append(ctx.timer_ns("holefill") * 1e-6)
    print("%-28s holefill %.4f ms (median of 200, back to back)" % (name, statistics.median(t)))
ctx.close()
