#!/usr/bin/env python3
"""Brick-skipping frame at the benchmark configuration in both schedules (sequential / RGBDR_FLAG_PIPELINE), with the
device-resident upload of bench.py's step: wall clock per frame."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
import torch  # noqa: E402
from rgbd_recon_amd import capi, synth  # noqa: E402

N, W, H = 4, 512, 424
dev = torch.device("cuda:0")
scene = synth.Scene(N, W, H, lut_res=(128, 106, 128), seed=1234)
c = capi.Context(capi.make_config(N, (W, H), voxel_size=2.0 / 512, brick_size=8 * 2.0 / 512), 0)
for i in range(N):
    c.set_calibration(i, scene.xyz[i], scene.lut_res, scene.uv[i], scene.lut_res, (0.5, 4.5))
    c.synth_inverse_calibration(i, scene.pinhole(i))
d_d = torch.from_numpy(scene.depth).to(dev)
d_c = torch.from_numpy(scene.color).to(dev)
torch.cuda.synchronize()
c.set_use_bricks(True)


def step():
    c.update_device(d_d.data_ptr(), d_c.data_ptr())
    c.clear_occupied_bricks(); c.process_textures(); c.update_occupied_bricks(); c.integrate()


out = {}
for pipelined in (False, True, False, True):
    c.set_pipelined(pipelined)
    for _ in range(10):
        step()
    c.sync()
    t0 = time.perf_counter()
    for _ in range(300):
        step()
    c.sync()
    out.setdefault("pipelined" if pipelined else "sequential", []).append(round((time.perf_counter() - t0) / 300 * 1e3, 4))
print(out)
c.close()
