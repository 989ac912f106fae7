#!/usr/bin/env python3
"""The reference's default operating point as bench.py's "reference_defaults" leg runs it (200 x 221 x 200 grid, 0.1 m
bricks, 0.007 m inverse LUTs generated on the device, DXT1 colour 1280 x 1080, brick-skipping sweep), alone, for
kernel traces: bash profiles/trace_probe.sh defaults profiles/defaults_probe.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
import torch  # noqa: E402
from rgbd_recon_amd import capi, synth  # noqa: E402

N, W, H = 4, 512, 424
dev = torch.device("cuda:0")
sc = synth.Scene(N, W, H, lut_res=(128, 106, 128), seed=1234, color_wh=(1280, 1080))
rc = capi.Context(capi.make_config(N, (W, H), color_wh=(1280, 1080), bbox_max=(1.0, 2.2, 1.0), voxel_size=0.01, brick_size=0.1,
                                   compress_rgb=1), 0)
for i in range(N):
    rc.set_calibration(i, sc.xyz[i], sc.lut_res, sc.uv[i], sc.lut_res, (0.5, 4.5))
    rc.set_inverse_calibration(i, rc.generate_inverse_lut(i, (286, 315, 286)), (286, 315, 286))
blocks = np.stack([synth.encode_dxt(sc.color[i], 1) for i in range(N)])
d_b = torch.from_numpy(np.ascontiguousarray(blocks)).to(dev)
d_d = torch.from_numpy(sc.depth).to(dev)
torch.cuda.synchronize()


def rstep():
    rc.update_device(d_d.data_ptr(), d_b.data_ptr())
    rc.clear_occupied_bricks(); rc.process_textures(); rc.update_occupied_bricks(); rc.integrate()


for _ in range(5):
    rstep()
rc.sync()
t0 = time.perf_counter()
for _ in range(100):
    rstep()
rc.sync()
print("ms per frame %.4f" % ((time.perf_counter() - t0) / 100 * 1e3))
rc.close()
