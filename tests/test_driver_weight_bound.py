"""The driver-tolerance study (oracle/driver_weight_bound.py): the oracle with every LINEAR weight
rounded to 8 fractional bits -- what NVIDIA's texture units do and GL 4.4 section 8.14 allows -- against the oracle with exact
weights.  Test infrastructure only: no parity test runs with the switch on, and the committed summary
(profiles/r06_driver_weight_bound.json) is what the small case reproduces here."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def test_the_switch_is_off_unless_the_study_sets_it(orc):
    assert orc.linear_weight_bits() == 0
    vol = np.arange(2 * 2 * 2 * 1, dtype=np.float32).reshape(2, 2, 2, 1)
    exact = orc.tex3d(vol, 0.4, 0.25, 0.25)              # t = 0.3: weight 0.3
    orc.set_linear_weight_bits(8)
    try:
        assert orc.linear_weight_bits() == 8
        coarse = orc.tex3d(vol, 0.4, 0.25, 0.25)         # weight 77 / 256
    finally:
        orc.set_linear_weight_bits(0)
    assert orc.linear_weight_bits() == 0
    # x weight 0.3 (then 77/256) between texels 0 and 1 of the row, y and z weights 0 (coordinate 0.25 of 2 texels = the centre of texel 0)
    assert abs(float(exact[0]) - 0.3) < 1e-6 and float(coarse[0]) == 77.0 / 256.0
    assert np.array_equal(orc.tex3d(vol, 0.4, 0.25, 0.25), exact)


def test_the_small_case_reproduces_the_committed_summary(orc):
    import driver_weight_bound as dwb
    want = json.load(open(os.path.join(ROOT, "profiles", "r06_driver_weight_bound.json")))["cases"]["small"]
    got = dwb.case_small()
    assert orc.linear_weight_bits() == 0
    for key in ("brick_counters", "occupied_list"):
        assert got[key] == want[key], key
    for key in ("voxels", "voxels_in_band", "voxels_beyond_5e-7", "voxels_changing_class", "max_abs_diff"):
        assert got["tsdf"][key] == want["tsdf"][key], key
    for img in ("depth_rg", "depth_b", "sil", "normal", "quality"):
        assert got[img]["values_differing"] == want[img]["values_differing"], img
    # what the study says, in one line each: the filtered depth flips validity at a handful of texels (the bounding-box test
    # of pre_depth.fs reads a LINEAR cv_xyz), each of which moves a voxel by a whole band; everything else stays tiny
    assert 0 < got["depth_rg"]["values_differing"] <= 8 and got["tsdf"]["voxels_changing_class"] <= 8
    assert got["tsdf"]["p99_abs_diff_in_band"] < 1e-6


def test_the_committed_study_covers_the_sizes_integration_md_quotes():
    j = json.load(open(os.path.join(ROOT, "profiles", "r06_driver_weight_bound.json")))
    assert set(j["cases"]) == {"small", "sample", "lut", "bands"}
    for name in ("sample", "lut", "bands"):
        t = j["cases"][name]["tsdf"]
        assert t["voxels_in_band"] > 50000 and 1e-5 < t["p99_abs_diff_in_band"] < 1e-3 and t["voxels_changing_class"] < 100
