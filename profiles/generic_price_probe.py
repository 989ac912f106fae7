#!/usr/bin/env python3
"""What RGBDR_FLAG_NO_RESAMPLE costs: the full sweep with the inverse LUT kept in its FILE layout (x-fastest RGBA32F, 8 taps per
voxel and sensor per frame: k_integrate_generic) against the default (the LUT resampled once to the grid's voxel centres, 12 B per
voxel and sensor streamed per frame: k_integrate_tiled), at
  ref   the reference's own operating point: grid 200 x 221 x 200, inverse LUT 286 x 315 x 286 (source/calib_inverter.cpp:10)
  512   a 512^3 grid with a 256^3 LUT (2 x coarser than the grid: the case the flag is for)
4 sensors 512 x 424.  Prints ms per sweep and the device memory the LUTs take; under rocprofv3 --pmc the kernels' HBM bytes
(profiles/pmc_generic.sh)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
import torch  # noqa: E402
from rgbd_recon_amd import capi, synth  # noqa: E402

N, W, H = 4, 512, 424
os.environ.setdefault("RGBDR_ARENA_TRIALS", "1")
scene = synth.Scene(N, W, H, lut_res=(128, 106, 128))
profiled = any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ)
for name, cfgkw, lut in (("ref", dict(bbox_max=(1.0, 2.2, 1.0), voxel_size=0.01, brick_size=0.1), (286, 315, 286)),
                         ("512", dict(voxel_size=2.0 / 512, brick_size=8 * 2.0 / 512), (256, 256, 256))):
    for flag_name, flags in (("resampled (default)", 15), ("RGBDR_FLAG_NO_RESAMPLE", 15 | 32)):
        free0 = torch.cuda.mem_get_info()[0]
        ctx = capi.Context(capi.make_config(N, (W, H), flags=flags, **cfgkw), 0)
        free1 = torch.cuda.mem_get_info()[0]
        for i in range(N):
            ctx.set_calibration(i, scene.xyz[i], scene.lut_res, scene.uv[i], scene.lut_res, (0.5, 4.5))
            ctx.set_inverse_calibration(i, ctx.generate_inverse_lut(i, lut), lut)
        ctx.sync()
        lut_bytes = free1 - torch.cuda.mem_get_info()[0]
        ctx.set_use_bricks(False)
        ctx.step(scene.depth, scene.color)
        ctx.enable_timer_accumulation(True)
        for _ in range(4 if profiled else 30):
            ctx.integrate()
        ns, n = ctx.timer_stats("2integrate")
        g = ctx.geo
        V = g.res_volume[0] * g.res_volume[1] * g.res_volume[2]
        print("%-4s grid %s LUT %s  %-24s sweep %.4f ms (%.1f Gvoxel/s)  LUT memory %.2f GiB (context without LUTs: %.2f GiB)"
              % (name, list(g.res_volume), list(lut), flag_name, ns / n * 1e-6, V / (ns / n) , lut_bytes / 2 ** 30, (free0 - free1) / 2 ** 30))
        ctx.close()
