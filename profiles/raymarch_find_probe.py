import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
load_package()
from rgbd_recon_amd import capi, synth
N, W, H, G = 4, 512, 424, 512
scene = synth.Scene(N, W, H, lut_res=(128, 106, 128))
ctx = capi.Context(capi.make_config(N, (W, H), voxel_size=2.0 / G, brick_size=8 * 2.0 / G), 0)
for i in range(N):
    ctx.set_calibration(i, scene.xyz[i], scene.lut_res, scene.uv[i], scene.lut_res, (0.5, 4.5))
    ctx.synth_inverse_calibration(i, scene.pinhole(i))
ctx.step(scene.depth, scene.color)
ctx.set_timer_detail(2); ctx.enable_timers(True)
view = capi.make_view((2.4, 1.8, 2.1), (0.0, 0.7, 0.0), (0.0, 1.0, 0.0), 45.0, 1280, 720, synth.BBOX_MIN, synth.BBOX_MAX)
for skip in (0, 1):
    view.skip_space = skip
    full, find = [], []
    for _ in range(15):
        ctx.raymarch(view); full.append(ctx.timer_ns("draw") * 1e-6)
        ctx.raymarch_find(view); find.append(ctx.timer_ns("draw") * 1e-6)
    print("skip %d: march + shade %.4f ms, march only (find) %.4f ms" % (skip, statistics.median(full), statistics.median(find)))
