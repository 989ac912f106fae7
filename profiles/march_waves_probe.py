#!/usr/bin/env python3
"""When does every wavefront of the whole-volume ray-march go through its stages?  Developer build of the library
(make -C rgbd-recon_amd/csrc trace): k_raymarch<0, 8> stamps s_memrealtime at entry, after the ray set-up, after the march and at exit
(kernels_raymarch.hip, MarchTrace).  The scene and view of profiles/display_frame_only.py; RGBDR_DISPLAY_GRID=ref|512."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
import torch  # noqa: E402
from rgbd_recon_amd import capi, synth  # noqa: E402

capi.LIB_PATH = os.path.join(ROOT, "profiles", "probes_src", "librgbdr_hip_trace.so")
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
view = capi.make_view((2.4, 1.8, 2.1), (0.0, 0.7, 0.0), (0.0, 1.0, 0.0), 45.0, 1280, 720, synth.BBOX_MIN, bmax)
view.skip_space = 1
for _ in range(20):
    rc.update_device(d_depth.data_ptr(), d_blocks.data_ptr())
    rc.clear_occupied_bricks()
    rc.process_textures()
    rc.update_occupied_bricks()
    rc.integrate()
    rc.draw(view, False)
rc.sync()
fn = capi.lib().rgbdr_debug_march_trace
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_size_t]
nw = (1280 // 16) * (720 // 16) * 4
buf = np.zeros((32768, 8), dtype=np.uint32)
assert fn(buf.ctypes.data, buf.nbytes) == 0
t = buf[:nw, :4].astype(np.int64)
t0 = t[:, 0].min()
us = (t - t0) / 100.0    # 100 MHz
rounds, whole = buf[:nw, 4], buf[:nw, 5]
start, setup, march, shade, end = us[:, 0], us[:, 1] - us[:, 0], us[:, 2] - us[:, 1], us[:, 3] - us[:, 2], us[:, 3]
print("%s grid: %d wavefronts, first entry to last exit %.1f us" % (which, nw, end.max()))
print("  entry times: median %.1f, 90 %% %.1f, last %.1f us" % (np.median(start), np.percentile(start, 90), start.max()))
work = rounds > 0
print("  wavefronts that march: %d; their set-up %.1f / march %.1f / shade %.1f us (medians); the others' whole stay %.1f us" %
      (work.sum(), np.median(setup[work]), np.median(march[work]), np.median(shade[work]), np.median((end - start)[~work])))
for name, v in (("set-up", setup), ("march", march), ("shade", shade)):
    print("  %-6s of the marching wavefronts: 50 %% %.1f  90 %% %.1f  99 %% %.1f  max %.1f us" %
          (name, np.percentile(v[work], 50), np.percentile(v[work], 90), np.percentile(v[work], 99), v[work].max()))
print("  the ten wavefronts that leave last: (entry, set-up, march, shade, exit us; rounds; rays taken by the whole wavefront)")
for k in np.argsort(-end)[:10]:
    print("    %7.1f %6.1f %6.1f %6.1f %7.1f  %3d %2d" % (start[k], setup[k], march[k], shade[k], end[k], rounds[k], whole[k]))
# busy wavefronts over time
edges = np.arange(0, end.max() + 10, 10)
busy = [(int(((start <= e) & (end > e)).sum()), int(((start <= e) & (end > e) & work).sum())) for e in edges]
print("  wavefronts in flight every 10 us (all, marching):", busy)
per_round = march[work] / np.maximum(rounds[work], 1)
print("  march time per round of the marching wavefronts: median %.2f us, 90 %% %.2f" % (np.median(per_round), np.percentile(per_round, 90)))
# k_depth_peels of the same frame: entry and exit of every wavefront
pt = buf[16384:16384 + nw][:, [0, 3]].astype(np.int64)
p0 = pt[:, 0].min()
ps, pe = (pt[:, 0] - p0) / 100.0, (pt[:, 1] - p0) / 100.0
stay = pe - ps
print("k_depth_peels: %d wavefronts, first entry to last exit %.1f us; entry times median %.1f, last %.1f us" % (nw, pe.max(), np.median(ps), ps.max()))
print("  a wavefront's stay: 50 %% %.1f  90 %% %.1f  99 %% %.1f  max %.1f us; summed %.0f wavefront-us = %.1f us on 2048 / %.1f us on 8192 slots" %
      (np.percentile(stay, 50), np.percentile(stay, 90), np.percentile(stay, 99), stay.max(), stay.sum(), stay.sum() / 2048, stay.sum() / 8192))
edges = np.arange(0, pe.max() + 5, 5)
print("  wavefronts in flight every 5 us:", [int(((ps <= e) & (pe > e)).sum()) for e in edges])
print("  the five that leave last (entry, stay, exit us):", [(round(float(ps[k]), 1), round(float(stay[k]), 1), round(float(pe[k]), 1)) for k in np.argsort(-pe)[:5]])
rc.close()
