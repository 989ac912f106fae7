#!/usr/bin/env python3
"""Host-fed frame rate (4 x 512 x 424 -> 256^3 and 512^3, full sweep): pageable upload / page-locked double buffer, sequential
and pipelined, interleaved rounds in one process."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
load_package()
import torch  # noqa
from rgbd_recon_amd import capi, synth
N, W, H = 4, 512, 424
G = int(sys.argv[1]) if len(sys.argv) > 1 else 512
scene = synth.Scene(N, W, H, lut_res=(128, 106, 128), seed=1234)
ctx = capi.Context(capi.make_config(N, (W, H), voxel_size=2.0 / G, brick_size=8 * 2.0 / G), 0)
for i in range(N):
    ctx.set_calibration(i, scene.xyz[i], scene.lut_res, scene.uv[i], scene.lut_res, (0.5, 4.5))
    ctx.synth_inverse_calibration(i, scene.pinhole(i))
ctx.set_use_bricks(False)
depth_h, color_h = np.ascontiguousarray(scene.depth), np.ascontiguousarray(scene.color)
d_depth = torch.from_numpy(scene.depth).cuda(); d_color = torch.from_numpy(scene.color).cuda()
def fed(upload, steps=60):
    for _ in range(3):
        upload(); ctx.clear_occupied_bricks(); ctx.process_textures(); ctx.update_occupied_bricks(); ctx.integrate()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        upload(); ctx.clear_occupied_bricks(); ctx.process_textures(); ctx.update_occupied_bricks(); ctx.integrate()
    ctx.sync()
    return round((time.perf_counter() - t0) / steps * 1e3, 4)
def mapped():
    ctx.map_frame_buffer(); ctx.upload_mapped_frame()
for _ in range(2):   # both page-locked buffers hold the frame set (a producer thread would have filled them)
    md, mc = ctx.map_frame_buffer(); md[:] = depth_h.view(np.uint8).reshape(-1); mc[:] = color_h.reshape(-1); ctx.upload_mapped_frame()
pre = sys.argv[2] if len(sys.argv) > 2 else ""
def steps(n):
    for _ in range(n):
        ctx.update_device(d_depth.data_ptr(), d_color.data_ptr()); ctx.clear_occupied_bricks(); ctx.process_textures(); ctx.update_occupied_bricks(); ctx.integrate()
    ctx.sync()
if "timers" in pre:
    ctx.set_timer_detail(0); ctx.enable_timer_accumulation(True); steps(20); ctx.timer_stats("2integrate"); ctx.enable_timer_accumulation(False); ctx.enable_timers(False); ctx.set_timer_detail(2)
if "bricks" in pre:
    ctx.set_use_bricks(True); steps(20); ctx.set_use_bricks(False)
if "pipe" in pre:
    ctx.set_pipelined(True); steps(20); ctx.set_pipelined(False)
if "elide" in pre:
    ctx.set_elide_stores(True); steps(20); ctx.set_elide_stores(False)
if "post" in pre:
    ctx.set_timer_detail(2); ctx.enable_timers(True)
    view = capi.make_view((2.4, 1.8, 2.1), (0.0, 0.7, 0.0), (0.0, 1.0, 0.0), 45.0, 1280, 720, synth.BBOX_MIN, synth.BBOX_MAX)
    ctx.raymarch(view); ctx.fill_colors(1280, 720); view.skip_space = 1; ctx.raymarch(view); ctx.enable_timers(False)
if "settle" in pre:
    ctx.settle(0.0)
out = {"pre": pre}
for rnd in range(2):
    for pipe in (False, True):
        ctx.set_pipelined(pipe)
        for name, up in (("device", lambda: ctx.update_device(d_depth.data_ptr(), d_color.data_ptr())),
                         ("pageable", lambda: ctx.update(depth_h, color_h)), ("mapped", mapped)):
            out.setdefault("%s/%s" % (name, "pipelined" if pipe else "sequential"), []).append(fed(up))
print(json.dumps(out))
