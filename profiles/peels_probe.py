#!/usr/bin/env python3
"""Depth peels + ray-march with space skipping at the benchmark configuration (4 sensors -> 512^3, 1280 x 720): timers."""
import os
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
import statistics  # noqa: E402
# busy = 1: twenty frames are queued in front of every draw, as in the frame loop (integrate, then draw) -- the GPU arrives
# at the view pass with its clocks up; busy = 0: draws back to back with the host's download of the frame in between
busy = int(os.environ.get("RGBDR_PROBE_BUSY", "1"))
import torch  # noqa: E402
dd, dc = torch.from_numpy(scene.depth).cuda(), torch.from_numpy(scene.color).cuda()
ctx.set_use_bricks(False)
for skip in (0, 1):
    view.skip_space = skip
    draw, peel = [], []
    for _ in range(25):
        for _k in range(20 if busy else 0):
            ctx.update_device(dd.data_ptr(), dc.data_ptr())
            ctx.clear_occupied_bricks(); ctx.process_textures(); ctx.update_occupied_bricks(); ctx.integrate()
        ctx.raymarch(view)
        draw.append(ctx.timer_ns("draw") * 1e-6)
        peel.append(ctx.timer_ns("brickdraw") * 1e-6 if skip else 0.0)
    # ("draw" brackets the ray-march kernel only; the depth peels that precede it with skip_space are "brickdraw")
    print("skip_space %d: draw (march) %.4f ms (median of 25; min %.4f), brickdraw (peels) %.4f ms, view pass %.4f ms"
          % (skip, statistics.median(draw), min(draw), statistics.median(peel), statistics.median(d + b for d, b in zip(draw, peel))))
ctx.close()
