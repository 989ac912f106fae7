#!/usr/bin/env python3
"""RGBDR_CU_SPLIT=n (the two-stream schedule with n CUs reserved for the pre_* chain): frame time and sweep time of the
benchmark configuration, sequential and pipelined.   python3 profiles/cu_split_probe.py [sensors] [grid]   (env decides n)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
load_package()
import torch  # noqa
from rgbd_recon_amd import capi, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
G = int(sys.argv[2]) if len(sys.argv) > 2 else 512
W, H = 512, 424
scene = synth.Scene(N, W, H, lut_res=(128, 106, 128), seed=1234)
ctx = capi.Context(capi.make_config(N, (W, H), voxel_size=2.0 / G, brick_size=8 * 2.0 / G), 0)
for i in range(N):
    ctx.set_calibration(i, scene.xyz[i], scene.lut_res, scene.uv[i], scene.lut_res, (0.5, 4.5))
    ctx.synth_inverse_calibration(i, scene.pinhole(i))
ctx.set_use_bricks(False)
d_depth = torch.from_numpy(scene.depth).cuda(); d_color = torch.from_numpy(scene.color).cuda()
def run(n):
    for _ in range(n):
        ctx.update_device(d_depth.data_ptr(), d_color.data_ptr()); ctx.clear_occupied_bricks(); ctx.process_textures(); ctx.update_occupied_bricks(); ctx.integrate()
    ctx.sync()
run(3); ctx.settle(2.0)
out = {"cu_split": os.environ.get("RGBDR_CU_SPLIT", "0"), "sensors": N, "grid": G}
for rnd in range(2):
    for pipe in (False, True):
        ctx.set_pipelined(pipe)
        run(5)
        ctx.set_timer_detail(0); ctx.enable_timer_accumulation(True)
        t0 = time.perf_counter(); run(60); dt = (time.perf_counter() - t0) / 60 * 1e3
        ns, n = ctx.timer_stats("2integrate")
        ctx.enable_timer_accumulation(False); ctx.enable_timers(False)
        out.setdefault("pipelined" if pipe else "sequential", []).append((round(dt, 4), round(ns / max(n, 1) * 1e-6, 4)))
print(json.dumps(out))
ctx.close()
