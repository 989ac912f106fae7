"""Developer probe: integrate launch time across re-created contexts in one
process (allocation placement) -- used to understand run-to-run variance."""
import sys, time, os
sys.path.insert(0, os.getcwd())
from __graft_entry__ import load_package
load_package()
# these probes drive the kernels' developer knobs (RGBDR_TILE_CHUNK, RGBDR_NT), which the shipped library does not read:
# build it with `make -C rgbd-recon_amd/csrc clean all EXTRA=-DRGBDR_DEV_KNOBS` first
if b"RGBDR_TILE_CHUNK" not in open(os.path.join(os.getcwd(), "rgbd-recon_amd", "librgbdr_hip.so"), "rb").read():
    raise SystemExit("librgbdr_hip.so was built without -DRGBDR_DEV_KNOBS: the knobs this probe sets would be ignored")
from rgbd_recon_amd import capi, synth
import torch, numpy as np
N,W,H,G=4,512,424,512
scene = synth.Scene(N, W, H, lut_res=(128,106,128))
def run(tag, pad_mb=0):
    pad = torch.empty(pad_mb*1024*1024, dtype=torch.uint8, device="cuda") if pad_mb else None
    ctx = capi.Context(capi.make_config(N,(W,H),voxel_size=2.0/G, brick_size=8*2.0/G), 0)
    for i in range(N):
        ctx.set_calibration(i, scene.xyz[i], scene.lut_res, scene.uv[i], scene.lut_res, (0.5,4.5))
        ctx.synth_inverse_calibration(i, scene.pinhole(i))
    ctx.step(scene.depth, scene.color)
    ctx.enable_timer_accumulation(True)
    out=[]
    ctx.set_use_bricks(BRICKS)
    for chunk in ("-1", "0", "64", "512", "-1", "64"):
        os.environ["RGBDR_TILE_CHUNK"] = chunk
        for _ in range(25):
            ctx.integrate()
        ns,n = ctx.timer_stats("2integrate")
        out.append((chunk, round(ns/n*1e-6,3)))
    v = ctx.device_tsdf()
    print(tag, pad_mb, out, hex(v.base))
    ctx.close()
    del pad
for BRICKS in (True, False):
    for k in range(2):
        run("bricks=%s %d" % (BRICKS, k))
