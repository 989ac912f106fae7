import os, sys
sys.path.insert(0, "/root/repo")
from __graft_entry__ import load_package
load_package()
import numpy as np
from rgbd_recon_amd import capi, synth
N, W, H, G = 4, 512, 424, 512
scene = synth.Scene(N, W, H, lut_res=(128, 106, 128))
ctx = capi.Context(capi.make_config(N, (W, H), voxel_size=2.0 / G, brick_size=8 * 2.0 / G), 0)
for i in range(N):
    ctx.set_calibration(i, scene.xyz[i], scene.lut_res, scene.uv[i], scene.lut_res, (0.5, 4.5))
    ctx.synth_inverse_calibration(i, scene.pinhole(i))
ctx.step(scene.depth, scene.color)
occ = ctx.get_occupied()[0].astype(np.int64)
rx, ry, rz = [int(v) for v in ctx.geo.res_bricks]
x, y, z = occ % rx, (occ // rx) % ry, occ // (rx * ry)
print("bricks", rx, ry, rz, "listed", len(occ), "x", x.min(), x.max(), "y", y.min(), y.max(), "z", z.min(), z.max())
for a, n in ((x, "x"), (y, "y"), (z, "z")):
    print(n, np.bincount(a, minlength=64).tolist())
