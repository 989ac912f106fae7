#!/usr/bin/env python3
"""Full sweep with RGBDR_FLAG_SKIP_BACKGROUND at the benchmark configuration (for kernel traces / timing)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
import torch  # noqa: E402,F401
from rgbd_recon_amd import capi, synth  # noqa: E402

if os.environ.get("RGBDR_PROBE_LIB"):       # A/B against a developer build under profiles/probes_src/
    capi.LIB_PATH = os.path.join(ROOT, os.environ["RGBDR_PROBE_LIB"])
N, W, H = 4, 512, 424
scene = synth.Scene(N, W, H, lut_res=(128, 106, 128), seed=1234)
c = capi.Context(capi.make_config(N, (W, H), voxel_size=2.0 / 512, brick_size=8 * 2.0 / 512), 0)
for i in range(N):
    c.set_calibration(i, scene.xyz[i], scene.lut_res, scene.uv[i], scene.lut_res, (0.5, 4.5))
    c.synth_inverse_calibration(i, scene.pinhole(i))
c.set_use_bricks(False)
c.update(scene.depth, scene.color)
for skip in (False, True):
    c.set_skip_background(skip)
    c.set_elide_stores(os.environ.get('SKIP_PROBE_ELIDE') == '1')
    for _ in range(5):
        c.clear_occupied_bricks(); c.process_textures(); c.update_occupied_bricks(); c.integrate()
    c.sync()
    t0 = time.perf_counter()
    for _ in range(40):
        c.clear_occupied_bricks(); c.process_textures(); c.update_occupied_bricks(); c.integrate()
    c.sync()
    print("skip", skip, "ms per frame %.4f" % ((time.perf_counter() - t0) / 40 * 1e3), c.skipped_pairs() if skip else "")
import numpy as np  # noqa: E402
v = c.readback_skip_tables(0)
org, dmin, dmax, ext = c.readback_skip_tables(1)
bounds = c.readback_skip_tables(2)
print("verdicts", {k: int((v == k).sum()) for k in range(4)}, "tiles with every sensor decided", int((v != 0).all(axis=1).sum()), "of", v.shape[0])
und = v == 0
live = und.sum(axis=1)
print("listed tiles by undecided sensors", {int(k): int((live == k).sum()) for k in range(1, N + 1)})
notcont = und & ~np.isfinite(dmin)
ox = (org & 0xffff).astype(np.int16).astype(int) + 1
oy = (org >> 16).astype(np.int16).astype(int) + 1
s_idx = np.broadcast_to(np.arange(N), v.shape)
bg = bounds[s_idx, ext, 0, oy, ox]
lo = bounds[s_idx, ext, 1, oy, ox]
hi = bounds[s_idx, ext, 2, oy, ox]
print('size classes', {k: int((ext == k).sum()) for k in range(3)})
mixed = und & np.isfinite(dmin) & ~np.isfinite(bg) & ~np.isfinite(hi)
surf = und & np.isfinite(dmin) & np.isfinite(hi)
bgw = und & np.isfinite(dmin) & np.isfinite(bg)
print("undecided: footprints leave window %d, mixed window %d, surface window (near the surface) %d, background window %d"
      % (notcont.sum(), mixed.sum(), surf.sum(), bgw.sum()))
if surf.any():
    print("surface windows: depth spread hi-lo median %.4f p90 %.4f (limit %.4f)" % (
        np.median((hi - lo)[surf]), np.percentile((hi - lo)[surf], 90), c.cfg.tsdf_limit))
c.close()
