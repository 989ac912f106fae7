#!/usr/bin/env python3
"""30 frames of the reference's default displayed frame (kinect_client.cpp:572-617: update, clear, processTextures,
updateOccupied, integrate, drawF; bricks on, skip-space on, colorfill on, DXT1 1280 x 1080 colour, 1280 x 720 window) through
rgbdr_draw: the workload of profiles/pmc_view_pass.sh.  RGBDR_DISPLAY_GRID=ref (default: 200 x 221 x 200, 10-voxel bricks,
inverse LUT 286 x 315 x 286) or 512 (the benchmark grid)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
import torch  # noqa: E402
from rgbd_recon_amd import capi, synth  # noqa: E402

if os.environ.get("RGBDR_PROBE_LIB"):   # A/B of two builds of the library in one session (profiles/ab_display.sh)
    capi.LIB_PATH = os.environ["RGBDR_PROBE_LIB"]
N, W, H = 4, 512, 424
which = os.environ.get("RGBDR_DISPLAY_GRID", "ref")
os.environ.setdefault("RGBDR_ARENA_TRIALS", "1")
sc = synth.Scene(N, W, H, lut_res=(128, 106, 128), seed=1234, color_wh=(1280, 1080))
blocks = np.stack([synth.encode_dxt(sc.color[i], 1) for i in range(N)])
d_depth, d_blocks = torch.from_numpy(sc.depth).cuda(), torch.from_numpy(np.ascontiguousarray(blocks)).cuda()
if which == "ref":
    bmax = (1.0, 2.2, 1.0)
    rc = capi.Context(capi.make_config(N, (W, H), color_wh=(1280, 1080), bbox_max=bmax, voxel_size=0.01, brick_size=0.1, compress_rgb=1), 0)
else:
    bmax = synth.BBOX_MAX
    rc = capi.Context(capi.make_config(N, (W, H), color_wh=(1280, 1080), voxel_size=2.0 / 512, brick_size=8 * 2.0 / 512, compress_rgb=1), 0)
for i in range(N):
    rc.set_calibration(i, sc.xyz[i], sc.lut_res, sc.uv[i], sc.lut_res, (0.5, 4.5))
    if which == "ref":
        rc.set_inverse_calibration(i, rc.generate_inverse_lut(i, (286, 315, 286)), (286, 315, 286))
    else:
        rc.synth_inverse_calibration(i, sc.pinhole(i))
if os.environ.get("RGBDR_DISPLAY_PIPELINE"):   # the pre_* chain of frame k + 1 on the second stream, under the view pass of frame k
    rc.set_pipelined(True)
view = capi.make_view((2.4, 1.8, 2.1), (0.0, 0.7, 0.0), (0.0, 1.0, 0.0), 45.0, 1280, 720, synth.BBOX_MIN, bmax)
view.skip_space = 1
import time  # noqa: E402


def frame():
    rc.update_device(d_depth.data_ptr(), d_blocks.data_ptr())
    rc.clear_occupied_bricks()
    rc.process_textures()
    rc.update_occupied_bricks()
    rc.integrate()
    rc.draw(view, True)


for _ in range(30):
    frame()
rc.sync()
if not any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):     # timings only without the profiler
    for _ in range(2000):
        frame()
    rc.sync()
    t0 = time.perf_counter()
    for _ in range(500):
        frame()
    host_ms = (time.perf_counter() - t0) / 500 * 1e3     # what the host needs to enqueue a frame (it runs ahead of the device)
    rc.sync()
    ms = (time.perf_counter() - t0) / 500 * 1e3
    t0 = time.perf_counter()
    for _ in range(20):     # ... and with the queue empty: the host's own cost per frame
        frame()
    host_idle_ms = (time.perf_counter() - t0) / 20 * 1e3
    rc.sync()
    rc.set_timer_detail(1)
    rc.enable_timers(True)
    frame()
    frame()
    rc.sync()
    st = {n: round(rc.timer_ns(t) * 1e-6, 4) for n, t in (("pre_chain", "1preprocess"), ("integrate", "2integrate"), ("depth_peels", "brickdraw"),
                                                            ("raymarch", "draw"), ("holefill", "holefill"), ("drawF", "3recon"))}
    print("%s grid %s: %.4f ms per displayed frame (500 back to back); stages of one more frame: %s; the host enqueues a frame in %.3f ms (%.3f into an empty queue)" %
          (which, list(rc.geo.res_volume), ms, st, host_ms, host_idle_ms))
elif os.environ.get("RGBDR_DISPLAY_SPLIT"):   # under the profiler: the march alone (k_raymarch<1,8>) and the shading alone (k_raymarch<2,1>)
    for _ in range(10):
        rc.raymarch_find(view)
        rc.raymarch_shade(view)
rc.close()
