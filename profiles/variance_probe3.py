"""Developer probe: tile order / non-temporal knobs per context (fast and slow placements)."""
import os, sys
sys.path.insert(0, os.getcwd())
from __graft_entry__ import load_package
load_package()
# these probes drive the kernels' developer knobs (RGBDR_TILE_CHUNK, RGBDR_NT), which the shipped library does not read:
# build it with `make -C rgbd-recon_amd/csrc clean all EXTRA=-DRGBDR_DEV_KNOBS` first
if b"RGBDR_TILE_CHUNK" not in open(os.path.join(os.getcwd(), "rgbd-recon_amd", "librgbdr_hip.so"), "rb").read():
    raise SystemExit("librgbdr_hip.so was built without -DRGBDR_DEV_KNOBS: the knobs this probe sets would be ignored")
from rgbd_recon_amd import capi, synth
N, W, H, G = 4, 512, 424, 512
scene = synth.Scene(N, W, H, lut_res=(128, 106, 128))
def make():
    ctx = capi.Context(capi.make_config(N, (W, H), voxel_size=2.0 / G, brick_size=8 * 2.0 / G), 0)
    for i in range(N):
        ctx.set_calibration(i, scene.xyz[i], scene.lut_res, scene.uv[i], scene.lut_res, (0.5, 4.5))
        ctx.synth_inverse_calibration(i, scene.pinhole(i))
    ctx.step(scene.depth, scene.color)
    ctx.set_use_bricks(False)
    ctx.enable_timer_accumulation(True)
    return ctx
def batch(ctx, n=20):
    for _ in range(3):
        ctx.integrate()
    ctx.sync(); ctx.timer_stats("2integrate")
    for _ in range(n):
        ctx.integrate()
    ctx.sync()
    ns, k = ctx.timer_stats("2integrate")
    return round(ns / k * 1e-6, 3)
ctxs = [make() for _ in range(int(sys.argv[1]))]
for k, c in enumerate(ctxs):
    row = {}
    for chunk in ("64", "0", "1", "8", "16", "512", "4096", "-1"):
        os.environ["RGBDR_TILE_CHUNK"] = chunk
        row["c" + chunk] = batch(c)
    os.environ["RGBDR_TILE_CHUNK"] = "64"
    os.environ["RGBDR_NT"] = "0"
    row["nt0"] = batch(c)
    del os.environ["RGBDR_NT"]
    print(k, row, flush=True)
