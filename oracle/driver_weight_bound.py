#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (oracle/): what a maintainer comparing against the authors' NVIDIA driver should expect.

Every LINEAR fetch of the path -- cv_xyz / cv_uv / cv_xyz_inv (framework/calibration/CalibVolumes.cpp:76,135,140), the
colour, quality and silhouette fetches (framework/NetKinectArray.cpp:46-53) -- is filtered by the texture unit.  GL 4.4
section 8.14 allows fixed-point weights and NVIDIA's units use them: the fraction of the texel coordinate is held with 8
fractional bits.  The committed fixtures come from Mesa llvmpipe (float weights for float textures), so that difference is
not in them.  This script DERIVES it: the oracle runs the same frames twice, with exact weights and with every LINEAR weight
rounded to 8 fractional bits (orc_set_linear_weight_bits), and reports what moves.

    python oracle/driver_weight_bound.py [--write profiles/r06_driver_weight_bound.json] [--cases small,sample,lut,bands]

Cases: `small` 2 x 128x106 into 64^3 (what tests/test_oracle_kat.py re-runs), `sample` BASELINE's sensor size, 4 x 512x424
into 128^3 with a 1:1 inverse LUT (a voxel centre hits its LUT texel with weight 0: only the 2-D fetches move), `lut` the same
frame with an inverse LUT 1.43 x finer than the grid (the reference's own ratio: 286x315x286 for 200x221x200,
source/calib_inverter.cpp:10), `bands` z bands of the 512^3 headline grid."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "golden")]
from __graft_entry__ import load_oracle, load_package  # noqa: E402


def classes(v, lim):
    return np.where(v <= -lim, -1, np.where(v >= lim, 1, 0))


def compare(a, b, limit):
    """a: exact weights, b: 8-bit weights -- the dict of run_pipeline / the banded volume"""
    lim = np.float32(limit)
    out = {}
    for k in ("depth_rg", "depth_b", "sil", "lab", "normal", "quality"):
        x, y = np.stack([np.asarray(v, np.float32) for v in a[k]]), np.stack([np.asarray(v, np.float32) for v in b[k]])
        ok = ~(np.isnan(x) | np.isnan(y))
        d = np.abs(x.astype(np.float64) - y)[ok]
        out[k] = {"max_abs_diff": float("%.3g" % (d.max() if d.size else 0.0)), "values_differing": int((x != y)[ok].sum()),
                  "values": int(ok.sum()), "nan_on_one_side_only": int((np.isnan(x) != np.isnan(y)).sum())}
    q0, q1 = np.stack(a["quality"]).astype(np.float64), np.stack(b["quality"]).astype(np.float64)
    ok = ~(np.isnan(q0) | np.isnan(q1)) & (q0 > 1e-6)
    out["quality"]["max_rel_diff_above_1e-6"] = float("%.3g" % (np.abs(q1 - q0)[ok] / q0[ok]).max()) if ok.any() else 0.0
    c0, c1 = a["counters"].astype(np.int64), b["counters"].astype(np.int64)
    out["brick_counters"] = {"sum": int(c0.sum()), "sum_abs_diff": int(np.abs(c0 - c1).sum()), "bricks_differing": int((c0 != c1).sum())}
    o0, o1 = set(np.asarray(a["occupied"]).tolist()), set(np.asarray(b["occupied"]).tolist())
    out["occupied_list"] = {"bricks": len(o0), "only_exact": len(o0 - o1), "only_8bit": len(o1 - o0)}
    t, r = np.asarray(a["tsdf"], np.float32).reshape(-1), np.asarray(b["tsdf"], np.float32).reshape(-1)
    ok = ~(np.isnan(t) | np.isnan(r))
    d = np.abs(t.astype(np.float64) - r)
    band = ok & ((np.abs(t) < lim) | (np.abs(r) < lim))
    flips = (classes(t, lim) != classes(r, lim)) & ok
    tie = (np.abs(np.abs(t) - lim) <= 1e-6) & (np.abs(np.abs(r) - lim) <= 1e-6) & (np.sign(t) == np.sign(r))
    dd = d[band]
    out["tsdf"] = {"voxels": int(ok.sum()), "voxels_in_band": int(band.sum()), "max_abs_diff": float("%.3g" % (d[ok].max() if ok.any() else 0.0)),
                   "max_abs_diff_over_limit": float("%.3g" % ((d[ok].max() if ok.any() else 0.0) / float(lim))),
                   "median_abs_diff_in_band": float("%.3g" % (np.median(dd) if dd.size else 0.0)),
                   "p99_abs_diff_in_band": float("%.3g" % (np.percentile(dd, 99) if dd.size else 0.0)),
                   "voxels_beyond_5e-7": int((d[ok] > 5e-7).sum()), "voxels_beyond_1e-5": int((d[ok] > 1e-5).sum()),
                   "voxels_beyond_1e-4": int((d[ok] > 1e-4).sum()),
                   "voxels_changing_class": int(flips.sum()), "voxels_changing_class_not_a_boundary_tie": int((flips & ~tie).sum()),
                   "nan_on_one_side_only": int((np.isnan(t) != np.isnan(r)).sum())}
    return out


def run_twice(fn, bits=8):
    orc = load_oracle()
    assert orc.linear_weight_bits() == 0
    a = fn()
    orc.set_linear_weight_bits(bits)
    try:
        b = fn()
    finally:
        orc.set_linear_weight_bits(0)
    return a, b


def case_small():
    pkg, orc = load_package(), load_oracle()
    from rgbd_recon_amd import capi, synth
    W, H, G = 128, 106, 64
    scene = synth.Scene(2, W, H, lut_res=(32, 27, 32), seed=1234)
    geo = capi.compute_geometry(capi.make_config(2, (W, H), voxel_size=2.0 / G, brick_size=8 * 2.0 / G))
    inv = scene.inverse((G, G, G))
    a, b = run_twice(lambda: orc.run_pipeline(scene, synth.BBOX_MIN, synth.BBOX_MAX, (G, G, G), inv, limit=0.01, brick_size=geo.brick_size,
                                              res_bricks=tuple(geo.res_bricks), use_bricks=False))
    return dict(compare(a, b, 0.01), what="2 sensors 128x106 into 64^3, 1:1 inverse LUT")


def sample_run(inv_res):
    load_package()
    orc = load_oracle()
    from rgbd_recon_amd import capi, synth
    G = 128
    scene = synth.Scene(4, 512, 424, lut_res=(32, 27, 32), seed=1234)
    cfg = capi.make_config(4, (512, 424), voxel_size=2.0 / G, brick_size=8 * 2.0 / G)
    geo = capi.compute_geometry(cfg)
    inv = scene.inverse(inv_res)
    return run_twice(lambda: orc.run_pipeline(scene, synth.BBOX_MIN, synth.BBOX_MAX, (G, G, G), inv, limit=cfg.tsdf_limit,
                                              brick_size=geo.brick_size, res_bricks=tuple(geo.res_bricks), min_voxels=cfg.min_voxels_per_brick,
                                              use_bricks=False)), cfg


def case_sample():
    (a, b), cfg = sample_run((128, 128, 128))
    return dict(compare(a, b, cfg.tsdf_limit), what="4 sensors 512x424 into 128^3, 1:1 inverse LUT (tests/golden/gl_sample_four_sensors_512x424_into_128)")


def case_lut():
    (a, b), cfg = sample_run((183, 183, 183))
    return dict(compare(a, b, cfg.tsdf_limit), what="4 sensors 512x424 into 128^3, inverse LUT 183^3 (1.43 x the grid, the reference's ratio): "
                "the cv_xyz_inv fetch of tsdf_integration.vs:31 has fractional weights too")


def case_bands():
    load_package()
    orc = load_oracle()
    from rgbd_recon_amd import synth
    import make_gl_golden as mg
    name = "four_sensors_512x424_into_512_bands"
    c = mg.BIG_SAMPLES[name]
    scene, cfg, geo, inv = mg.big_scene(name)
    G = c["G"]

    def once():
        ref = orc.run_pipeline(scene, synth.BBOX_MIN, synth.BBOX_MAX, (G, G, G), None, limit=cfg.tsdf_limit, brick_size=geo.brick_size,
                               res_bricks=tuple(geo.res_bricks), min_voxels=cfg.min_voxels_per_brick, use_bricks=False)
        vols = []
        for z0, z1 in c["bands"]:
            luts = []
            for a in inv:
                full = np.zeros((G, G, G, 4), np.float32)
                full[z0:z1, ..., :a.shape[-1]] = a[z0:z1]
                luts.append(full)
            vol = np.full((G, G, G), np.nan, np.float32)
            orc.integrate(luts, ref["sil"], ref["depth_b"], ref["quality"], (G, G, G), cfg.tsdf_limit, z_range=(z0, z1), out=vol)
            vols.append(vol[z0:z1].copy())
            del vol, luts
        ref["tsdf"] = np.concatenate(vols)
        return ref
    a, b = run_twice(once)
    return dict(compare(a, b, cfg.tsdf_limit), what="4 sensors 512x424 into z bands %s of the 512^3 headline grid, 1:1 inverse LUT" % (c["bands"],))


CASES = {"small": case_small, "sample": case_sample, "lut": case_lut, "bands": case_bands}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="small,sample,lut,bands")
    ap.add_argument("--write", default=None)
    args = ap.parse_args()
    res = {"weights": "every LINEAR weight rounded to 8 fractional bits (round to nearest) against exact binary32 weights, same oracle, same frames",
           "cases": {}}
    for k in args.cases.split(","):
        res["cases"][k] = CASES[k]()
        print(k, json.dumps(res["cases"][k]), flush=True)
    if args.write:
        with open(args.write, "w") as f:
            json.dump(res, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
