"""Developer probe: time of the on-device inverse-LUT generation at benchmark scale."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
load_package()
from rgbd_recon_amd import capi, synth
import numpy as np
N, W, H = 1, 512, 424
scene = synth.Scene(N, W, H, lut_res=(128, 106, 128))
for G in (256, 512):
    ctx = capi.Context(capi.make_config(N, (W, H), voxel_size=2.0 / G, brick_size=8 * 2.0 / G), 0)
    ctx.set_calibration(0, scene.xyz[0], scene.lut_res, scene.uv[0], scene.lut_res, (0.5, 4.5))
    for R in (2, 3):
        ctx.compute_inverse_calibration(0, R)
        ctx.sync(); t0 = time.perf_counter()
        ctx.compute_inverse_calibration(0, R)
        ctx.sync(); dt = time.perf_counter() - t0
        print("  first window %d: widened %d, exhaustive %d of %d voxels" % ((R,) + ctx.inverse_search_stats(0) + (G ** 3,)))
        inv = ctx.readback_inverse_calibration(0, G // 2, G // 2 + 2)
        ana = synth.inverse_lut(scene.sensors[0], (G, G, G), z_range=(G // 2, G // 2 + 2))
        both = (ana[..., 3] > 0) & (inv[..., 0] >= 0)
        print("grid %d^3 window %d: %.3f s (%.1f Mvoxel/s), mean |d(u,v,d)| vs analytic %.5f" %
              (G, R, dt, G ** 3 / dt / 1e6, np.abs(ana[both][:, :3] - inv[both][:, :3]).mean()))
    ctx.close()
