import os, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
load_package()
from rgbd_recon_amd import capi, synth
N, W, H = 4, 512, 424
os.environ.setdefault("RGBDR_ARENA_TRIALS", "1")
sc = synth.Scene(N, W, H, lut_res=(128, 106, 128), seed=1234)
for G in (512,):
    rc = capi.Context(capi.make_config(N, (W, H), voxel_size=2.0 / G, brick_size=8 * 2.0 / G), 0)
    for i in range(N):
        rc.set_calibration(i, sc.xyz[i], sc.lut_res, sc.uv[i], sc.lut_res, (0.5, 4.5))
        rc.synth_inverse_calibration(i, sc.pinhole(i))
    rc.step(sc.depth, sc.color)
    view = capi.make_view((2.4, 1.8, 2.1), (0.0, 0.7, 0.0), (0.0, 1.0, 0.0), 45.0, 1280, 720, synth.BBOX_MIN, synth.BBOX_MAX)
    view.skip_space = 1
    col, dep, ns = rc.raymarch(view)
    num = np.rint(ns / 0.0027).astype(np.int64)
    hit = dep < 1
    print("pixels", num.size, "hit", hit.mean(), "zero-sample", (num == 0).mean())
    print("samples: total %.3g, by hit rays %.3g, by miss rays %.3g" % (num.sum(), num[hit].sum(), num[~hit].sum()))
    for q in (50, 90, 99, 99.9, 100):
        print("  percentile %5.1f of nonzero rays: %d samples (hit rays: %d, miss rays: %d)" % (q, np.percentile(num[num > 0], q), np.percentile(num[hit], q), np.percentile(num[(~hit) & (num > 0)], q) if ((~hit) & (num > 0)).any() else 0))
    # per 8x8 wave square: max samples in the square (what the wavefront runs for)
    sq = num[:720, :1280].reshape(90, 8, 160, 8).max(axis=(1, 3))
    print("wave squares: %d, with work %d; sum of per-square max %.3g; max %d" % (sq.size, (sq > 0).sum(), sq.sum(), sq.max()))
    rc.close()
