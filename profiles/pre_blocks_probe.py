#!/usr/bin/env python3
"""Where and when the wavefronts of the two 13 x 13 kernels of the pre_* chain run (4 x 512 x 424 benchmark frames).
Needs the developer build of the library:
    make -C rgbd-recon_amd/csrc trace      (-> profiles/probes_src/librgbdr_hip_trace.so; the product library is not touched)
    python3 profiles/pre_blocks_probe.py [out.json]
Per kernel: heavy wavefronts per SIMD / CU / XCD (heavy = ran the tap loops), when the last wavefront of each CU
finished, and the span of the launch by the wavefronts' own clocks."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
import torch  # noqa: E402,F401
from rgbd_recon_amd import capi, synth  # noqa: E402

capi.LIB_PATH = os.path.join(ROOT, "profiles", "probes_src", "librgbdr_hip_trace.so")
N, W, H, G = 4, 512, 424, 128
scene = synth.Scene(N, W, H, lut_res=(128, 106, 128), seed=1234)
ctx = capi.Context(capi.make_config(N, (W, H), voxel_size=2.0 / G, brick_size=8 * 2.0 / G), 0)
for i in range(N):
    ctx.set_calibration(i, scene.xyz[i], scene.lut_res, scene.uv[i], scene.lut_res, (0.5, 4.5))
ctx.update(scene.depth, scene.color)
for _ in range(30):
    ctx.clear_occupied_bricks()
    ctx.process_textures()
    ctx.update_occupied_bricks()
ctx.sync()
trace = capi.lib().rgbdr_debug_wave_trace
trace.restype = C.c_int
trace.argtypes = [C.c_int, C.c_void_p, C.c_size_t]
gx, gy = (W // 16) * N, (H + 15) // 16
nw = gx * gy * 4
out = {}
for which, name in ((0, "k_pre_depth"), (1, "k_boundary_normal_quality")):
    buf = np.zeros((4096 * 4, 6), np.uint32)
    assert trace(which, buf.ctypes.data, buf.nbytes) == 0
    t = buf[:nw]
    hw, xcc, t0, t1, clk, flag = (t[:, k].astype(np.int64) for k in range(6))
    # HW_ID (gfx9): wave_id [3:0], simd_id [5:4], pipe_id [7:6], cu_id [11:8], sh_id [12], se_id [15:13]
    simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
    xc = xcc & 15
    cu_key = ((xc * 8 + se) * 2 + sh) * 16 + cu
    simd_key = cu_key * 4 + simd
    dur_us = (t1 - t0) / 100.0
    start = t0.min()
    end_us = (t1 - start) / 100.0
    beg_us = (t0 - start) / 100.0
    heavy = dur_us > 0.35 * dur_us.max()
    res = {"wavefronts": int(nw), "launch_span_us": float(end_us.max()), "heavy_wavefronts": int(heavy.sum()),
           "heavy_by_flag": int((flag >= 1).sum()) if which == 0 else None,
           "heavy_duration_us": {"median": float(np.median(dur_us[heavy])), "p90": float(np.percentile(dur_us[heavy], 90)), "max": float(dur_us.max())},
           "light_duration_us_median": float(np.median(dur_us[~heavy])),
           "last_start_us": float(beg_us.max()), "heavy_last_start_us": float(beg_us[heavy].max()),
           "distinct": {"xcc": int(len(set(xc))), "cu": int(len(set(cu_key))), "simd": int(len(set(simd_key)))}}
    for label, key in (("simd", simd_key), ("cu", cu_key), ("xcc", xc)):
        ids = np.unique(key)
        cnt = np.array([int(heavy[key == i].sum()) for i in ids])
        busy = np.array([float(dur_us[(key == i) & heavy].sum()) for i in ids])
        fin = np.array([float(end_us[key == i].max()) for i in ids])
        res["heavy_per_" + label] = {"mean": float(cnt.mean()), "max": int(cnt.max()), "min": int(cnt.min()),
                                     "hist": np.bincount(cnt).tolist() if label != "xcc" else cnt.tolist()}
        res["finish_us_per_" + label] = {"mean": float(fin.mean()), "max": float(fin.max()), "p10": float(np.percentile(fin, 10))}
        if label == "simd":
            res["heavy_busy_us_per_simd"] = {"mean": float(busy.mean()), "max": float(busy.max())}
    # does the heavy count of a SIMD explain when it finishes?
    ids = np.unique(simd_key)
    cnt = np.array([int(heavy[simd_key == i].sum()) for i in ids])
    fin = np.array([float(end_us[simd_key == i].max()) for i in ids])
    res["finish_us_by_heavy_count"] = {int(c): float(fin[cnt == c].mean()) for c in np.unique(cnt)}
    # image position of the heavy blocks of the busiest CU
    blk = np.arange(nw) // 4
    bx, by = blk % gx, blk // gx
    ids = np.unique(cu_key)
    cntc = np.array([int(heavy[cu_key == i].sum()) for i in ids])
    worst = ids[int(np.argmax(cntc))]
    sel = (cu_key == worst) & heavy
    res["busiest_cu_blocks_xy"] = sorted(set(zip(bx[sel].tolist(), by[sel].tolist())))[:40]
    out[name] = res
    print(name, json.dumps(res))
# ---- experiment: launch orders built from the measured heavy blocks (an oracle's knowledge of the frame) ----
import time

setter = capi.lib().rgbdr_debug_block_order
setter.restype = C.c_int
setter.argtypes = [C.c_int, C.c_void_p, C.c_size_t]
nb = gx * gy


def block_heavy(which):
    buf = np.zeros((4096 * 4, 6), np.uint32)
    assert trace(which, buf.ctypes.data, buf.nbytes) == 0
    d = (buf[:nw, 3].astype(np.int64) - buf[:nw, 2].astype(np.int64)).reshape(nb, 4).max(axis=1)
    return d > 0.35 * d.max()


def chain_ms(frames=400):
    for _ in range(20):
        ctx.clear_occupied_bricks(); ctx.process_textures(); ctx.update_occupied_bricks()
    ctx.sync()
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        for _ in range(frames):
            ctx.clear_occupied_bricks(); ctx.process_textures(); ctx.update_occupied_bricks()
        ctx.sync()
        best = min(best, (time.perf_counter() - t0) / frames * 1e3)
    return best


def span(which):
    buf = np.zeros((4096 * 4, 6), np.uint32)
    assert trace(which, buf.ctypes.data, buf.nbytes) == 0
    t = buf[:nw].astype(np.int64)
    return float((t[:, 3].max() - t[:, 2].min()) / 100.0)


heavy = [block_heavy(0), block_heavy(1)]
orders = {}
ids = np.arange(nb)
for which in (0, 1):
    h = heavy[which]
    cls = np.empty(nb, np.int64)
    for g in range(8):
        m = ids[g::8]
        cls[g::8] = np.concatenate([m[h[m]], m[~h[m]]])
    orders.setdefault("heavy first inside each XCD class", []).append(cls)
    orders.setdefault("heavy first, all classes together", []).append(np.concatenate([ids[h], ids[~h]]))
    # rows from the centre outwards (no knowledge of the frame at all)
    rows = sorted(range(gy), key=lambda r: abs(r - (gy - 1) / 2.0))
    orders.setdefault("rows centre-out", []).append(np.concatenate([np.arange(r * gx, (r + 1) * gx) for r in rows]))
    rev = np.empty(nb, np.int64)
    for g in range(8):
        m = ids[g::8]
        rev[g::8] = np.concatenate([m[~h[m]], m[h[m]]])
    orders.setdefault("heavy LAST inside each XCD class", []).append(rev)
exp = {"natural": {"chain_ms": chain_ms(), "span_us": [span(0), span(1)]}}
for name, (o0, o1) in orders.items():
    for which, o in ((0, o0), (1, o1)):
        assert sorted(o.tolist()) == list(range(nb))
        a = np.ascontiguousarray(o, dtype=np.uint16)
        assert setter(which, a.ctypes.data, a.size) == 0
    exp[name] = {"chain_ms": chain_ms(), "span_us": [span(0), span(1)]}
    for which in (0, 1):
        assert setter(which, None, 0) == 0
exp["natural again"] = {"chain_ms": chain_ms(), "span_us": [span(0), span(1)]}
out["launch_order_experiment"] = exp
print(json.dumps(exp, indent=1))
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
ctx.close()
