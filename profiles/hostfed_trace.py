#!/usr/bin/env python3
"""Host-fed frames (4 x 512 x 424 -> 512^3, full sweep) through the page-locked double frame buffer, one schedule per run:
    python3 profiles/hostfed_trace.py sequential|pipelined|device [frames]
The workload of `rocprofv3 --kernel-trace --memory-copy-trace` in profiles/hostfed_trace.sh; prints ms per frame."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
load_package()
import torch  # noqa
from rgbd_recon_amd import capi, synth
mode = sys.argv[1] if len(sys.argv) > 1 else "sequential"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 40
N, W, H, G = 4, 512, 424, 512
scene = synth.Scene(N, W, H, lut_res=(128, 106, 128), seed=1234)
ctx = capi.Context(capi.make_config(N, (W, H), voxel_size=2.0 / G, brick_size=8 * 2.0 / G), 0)
for i in range(N):
    ctx.set_calibration(i, scene.xyz[i], scene.lut_res, scene.uv[i], scene.lut_res, (0.5, 4.5))
    ctx.synth_inverse_calibration(i, scene.pinhole(i))
ctx.set_use_bricks(False)
d_depth = torch.from_numpy(scene.depth).cuda(); d_color = torch.from_numpy(scene.color).cuda()
depth_h, color_h = np_d, np_c = scene.depth, scene.color
def mapped():
    ctx.map_frame_buffer(); ctx.upload_mapped_frame()
up = {"device": lambda: ctx.update_device(d_depth.data_ptr(), d_color.data_ptr()), "pageable": lambda: ctx.update(depth_h, color_h)}.get(mode.split("-")[0], mapped)
ctx.set_pipelined("pipelined" in mode)
for _ in range(2):            # both page-locked buffers hold the frame set ("the producer filled it on its own thread")
    md, mc = ctx.map_frame_buffer()
    md[:] = scene.depth.view("uint8").reshape(-1); mc[:] = scene.color.reshape(-1)
    ctx.upload_mapped_frame()
def run(n):
    for _ in range(n):
        up(); ctx.clear_occupied_bricks(); ctx.process_textures(); ctx.update_occupied_bricks(); ctx.integrate()
    ctx.sync()
run(5)
t0 = time.perf_counter(); run(frames); dt = (time.perf_counter() - t0) / frames * 1e3
print(json.dumps({"mode": mode, "frames": frames, "ms_per_frame": round(dt, 4)}))
ctx.close()
