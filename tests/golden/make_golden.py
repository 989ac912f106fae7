#!/usr/bin/env python3
"""Generates tests/golden/*.npz.

The reference ships no golden vectors, recordings or calibration files
(SURVEY.md section 4), and its GLSL cannot run in this container, so these
fixtures are produced by the CPU oracle (oracle/rgbdr_oracle.c) on the seeded
synthetic scene: they pin the oracle against regressions and give the GPU tests a
fixed target that does not depend on the oracle being rebuilt identically.  They
are NOT outputs of the reference ("parity unpinned").

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_oracle, load_package  # noqa: E402

CASES = {
    # name: sensors, (W, H), forward LUT res, TSDF grid, inverse LUT res, flags
    "two_sensors_1to1": dict(n=2, wh=(64, 53), lut=(16, 13, 16), G=32, inv=(32, 32, 32), flags=15),
    "two_sensors_generic": dict(n=2, wh=(64, 53), lut=(16, 13, 16), G=32, inv=(22, 22, 22), flags=15),
    "three_sensors_nobricks": dict(n=3, wh=(48, 40), lut=(12, 10, 12), G=24, inv=(24, 24, 24), flags=7),
}


def inputs(case):
    load_package()
    from rgbd_recon_amd import synth

    scene = synth.Scene(case["n"], case["wh"][0], case["wh"][1], lut_res=case["lut"], seed=4321)
    inv = scene.inverse(case["inv"])
    return scene, inv, synth


def run(case):
    orc = load_oracle()
    scene, inv, synth = inputs(case)
    G = case["G"]
    f = case["flags"]
    voxel = np.float32(2.0 / G)
    brick = orc.adjust_brick_size(float(8 * voxel), float(voxel))
    rb = orc.divide_box(synth.BBOX_MIN, synth.BBOX_MAX, brick)
    ref = orc.run_pipeline(scene, synth.BBOX_MIN, synth.BBOX_MAX, (G, G, G), inv, limit=0.01, brick_size=brick, bv=8,
                           res_bricks=rb, filter_textures=bool(f & 1), processed=bool(f & 2), refine=bool(f & 4),
                           use_bricks=bool(f & 8))
    out = {"depth": scene.depth, "color": scene.color, "counters": ref["counters"], "occupied": ref["occupied"],
           "tsdf": ref["tsdf"]}
    for i in range(case["n"]):
        out["xyz%d" % i] = scene.xyz[i]
        out["uv%d" % i] = scene.uv[i]
        out["inv%d" % i] = inv[i]
        for k in ("morph", "depth_rg", "lab", "depth_b", "sil", "normal", "quality"):
            out["%s%d" % (k, i)] = ref[k][i]
    return out


if __name__ == "__main__":
    for name, case in CASES.items():
        out = run(case)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print(name, "%.0f KiB" % (os.path.getsize(path) / 1024))
