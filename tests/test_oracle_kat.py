"""Known-answer tests of the CPU oracle: the reference has
no tests or golden data, so the restatement is checked against analytic cases
derived from the shader source."""
import math

import numpy as np

LIMIT = 0.01


def identity_inverse(G):
    """inverse LUT that maps volume position p to itself: (x, y, z, 1)"""
    c = (np.arange(G) + 0.5) / G
    Z, Y, X = np.meshgrid(c, c, c, indexing="ij")
    return np.stack([X, Y, Z, np.ones_like(X)], axis=-1).astype(np.float32)


def plane_frame(W, H, z0, q=1.0):
    sil = np.ones((H, W), np.float32)
    db = np.zeros((H, W, 2), np.float32)
    db[..., 0] = z0
    qual = np.full((H, W), q, np.float32)
    return sil, db, qual


# ---- sampling ---------------------------------------------------------------
def test_linear_sampling_at_texel_centres_and_clamp(orc):
    rng = np.random.default_rng(0)
    vol = rng.standard_normal((4, 5, 6, 3)).astype(np.float32)
    for (x, y, z) in [(0, 0, 0), (5, 4, 3), (2, 3, 1)]:
        got = orc.tex3d(vol, (x + 0.5) / 6, (y + 0.5) / 5, (z + 0.5) / 4)
        assert np.array_equal(got, vol[z, y, x])
    # CLAMP_TO_EDGE: outside [0,1] returns the edge texel
    assert np.array_equal(orc.tex3d(vol, -3.0, 0.5 / 5, 0.5 / 4), vol[0, 0, 0])
    assert np.array_equal(orc.tex3d(vol, 7.0, 4.5 / 5, 3.5 / 4), vol[3, 4, 5])
    # midway between two texels along x
    got = orc.tex3d(vol, 1.0 / 6, 0.5 / 5, 0.5 / 4)
    exp = vol[0, 0, 0] + np.float32(0.5) * (vol[0, 0, 1] - vol[0, 0, 0])
    assert np.array_equal(got, exp.astype(np.float32))


def test_linear_of_constant_is_exact(orc):
    # the blend T0 + a*(T1-T0) returns the constant exactly: silhouette < 1.0 must
    # not fire inside an all-ones silhouette (tsdf_integration.vs:33)
    img = np.ones((9, 11), np.float32)
    rng = np.random.default_rng(1)
    for _ in range(500):
        assert orc.tex2d(img, rng.uniform(-0.2, 1.2), rng.uniform(-0.2, 1.2))[0] == 1.0


def test_nearest_and_nan(orc):
    l = orc.lib()
    assert l.orc_axis_nearest(0.0, 8) == 0
    assert l.orc_axis_nearest(0.999, 8) == 7
    assert l.orc_axis_nearest(1.0, 8) == 7
    assert l.orc_axis_nearest(-5.0, 8) == 0
    assert l.orc_axis_nearest(3.5 / 8, 8) == 3
    assert l.orc_axis_nearest(float("nan"), 8) == 0
    assert l.orc_axis_nearest(1e30, 8) == 7
    img = np.arange(12, dtype=np.float32).reshape(3, 4)
    assert math.isnan(orc.tex2d(img, float("nan"), 0.5)[0])


# ---- pre_morph --------------------------------------------------------------
def test_morph_keeps_valid_fills_holes(orc):
    d = np.full((7, 7), 2.0, np.float32)
    d[3, 3] = 0.0           # hole with 8 valid neighbours -> their mean
    d[0, 0] = 5.0           # out of range (>= 4.5), neighbours valid
    out = orc.morph(d, 0)
    assert out[3, 3] == 2.0 and out[0, 0] == 2.0
    assert np.array_equal(out[1:3, 1:3], d[1:3, 1:3])
    e = np.zeros((5, 5), np.float32)
    assert np.array_equal(orc.morph(e, 0), e)          # nothing valid -> 0
    assert np.array_equal(orc.morph(d, 1), d)          # mode 1 = copy (pre_morph.fs:130-131)
    # is_valid is 0.5 < d < 4.5, both strict (:36-39): exactly 0.5 m and 4.5 m are holes and get filled
    for edge in (0.5, 4.5):
        d = np.full((5, 5), 2.0, np.float32)
        d[2, 2] = edge
        assert orc.morph(d, 0)[2, 2] == 2.0


def test_morph_second_mean_rejects_outliers(orc):
    d = np.zeros((3, 3), np.float32)
    d[0, :] = [1.0, 1.0, 1.1]
    d[2, 0] = 3.0           # far from the first mean (1.525): dropped in the second pass
    out = orc.morph(d, 0)
    first = np.float32((np.float32(1.0) + np.float32(1.0) + np.float32(1.1) + np.float32(3.0)) / np.float32(4.0))
    assert abs(first - 1.525) < 1e-6
    # second pass keeps |mean - d| < 0.2: none of 1.0/1.0/1.1/3.0 -> 0
    assert out[1, 1] == 0.0


# ---- Lab --------------------------------------------------------------------
def test_lab_linear_branch_with_reference_quirk(orc):
    # rgb_to_xyz divides an already normalised colour by 255 (inc_color.glsl:14-16),
    # so both pow() branches are unreachable for [0,1] input: Lab is affine
    rgb = np.array([0.2, 0.5, 0.9], np.float64)
    lin = rgb / 255.0 / 12.92 * 100.0
    X = lin @ [0.4124, 0.3576, 0.1805]
    Y = lin @ [0.2126, 0.7152, 0.0722]
    Z = lin @ [0.0193, 0.1192, 0.9505]
    f = lambda n: (903.3 * n + 16.0) / 116.0
    x, y, z = f(X / 95.047), f(Y / 100.0), f(Z / 108.883)
    exp = [max(0.0, 116 * y - 16), 500 * (x - y), 200 * (y - z)]
    np.testing.assert_allclose(orc.rgb_to_lab(rgb), exp, rtol=1e-4, atol=2e-6)


def test_pow_products_match_libm(orc):
    rng = np.random.default_rng(2)
    for x in rng.uniform(0, 1, 200).astype(np.float32):
        exact = float(x) ** 6
        assert abs(orc.lib().orc_pow6(x) - exact) <= 3 * np.spacing(np.float32(exact))
    assert orc.lib().orc_pow2(-0.5) == 0.25     # IEEE-like for negative base


# ---- pre_boundary -----------------------------------------------------------
def test_boundary_classes(orc):
    H, W = 9, 9
    rg = np.zeros((H, W, 2), np.float32)
    rg[..., 0] = 0.4
    rg[..., 1] = 0.9                     # confident interior
    rg[0, 0] = (0.0, 0.0)                # outside box
    rg[4, 4] = (0.4, 0.3)                # low range-quality, uniform colour -> refined edge
    lab = np.zeros((H, W, 3), np.float32)
    db, sil = orc.boundary(rg, lab, refine=True)
    assert tuple(db[2, 2]) == (np.float32(0.4), 0.0) and sil[2, 2] == 1.0
    assert tuple(db[0, 0]) == (0.0, 0.0) and sil[0, 0] == 0.0
    assert tuple(db[4, 4]) == (np.float32(0.4), 1.0) and sil[4, 4] == 0.0   # silhouette stays 0 (:103-112)
    db2, sil2 = orc.boundary(rg, lab, refine=False)
    assert tuple(db2[4, 4]) == (-1.0, np.float32(0.1)) and sil2[4, 4] == 0.0
    lab2 = lab.copy()
    lab2[4, 4] = (5.0, 0.0, 0.0)         # colour differs from the neighbourhood by > 0.5
    db3, _ = orc.boundary(rg, lab2, refine=True)
    assert tuple(db3[4, 4]) == (-1.0, np.float32(0.1))
    # the confidence threshold is 0.65 (pre_boundary.fs:102): 0.66 is interior, 0.64 goes through the edge test
    rg5 = rg.copy()
    rg5[4, 4] = (0.4, 0.66)
    rg5[6, 6] = (0.4, 0.64)
    db5, sil5 = orc.boundary(rg5, lab, refine=True)
    assert tuple(db5[4, 4]) == (np.float32(0.4), 0.0) and sil5[4, 4] == 1.0
    assert tuple(db5[6, 6]) == (np.float32(0.4), 1.0) and sil5[6, 6] == 0.0
    rg4 = rg.copy()
    rg4[2:7, 2:7, 1] = 0.3               # fewer than 8 confident neighbours -> colour distance 1
    db4, _ = orc.boundary(rg4, lab, refine=True)
    assert tuple(db4[4, 4]) == (-1.0, np.float32(0.1))
    # num_samples < total_samples * 0.5 = 8 (pre_boundary.fs:53): 7 confident neighbours of identical colour are not enough, 8 are
    rg7 = rg4.copy()
    rg7[2, 2:7, 1] = 0.9
    rg7[3, 2:4, 1] = 0.9
    assert tuple(orc.boundary(rg7, lab, refine=True)[0][4, 4]) == (-1.0, np.float32(0.1))
    rg7[3, 4, 1] = 0.9
    assert tuple(orc.boundary(rg7, lab, refine=True)[0][4, 4]) == (np.float32(0.4), 1.0)


# ---- integration (SURVEY 8c i-iv) ------------------------------------------
def test_single_plane_piecewise(orc):
    G, W, H, z0 = 16, 8, 8, 0.5
    inv = identity_inverse(G)
    sil, db, q = plane_frame(W, H, z0)
    t = orc.integrate([inv], [sil], [db], [q], (G, G, G), LIMIT)
    zc = ((np.arange(G) + np.float32(0.5)) * np.float32(1.0 / G)).astype(np.float32)
    s = zc - np.float32(z0)
    exp = np.where(s <= -LIMIT, -LIMIT, np.where(s >= LIMIT, LIMIT, s)).astype(np.float32)
    assert np.array_equal(t, np.broadcast_to(exp[:, None, None], t.shape))


def test_in_band_value_is_signed_distance(orc):
    G = 64
    z0 = (31 + 0.5) / G + 0.004          # surface 0.004 behind the centre of layer 31
    inv = identity_inverse(G)
    sil, db, q = plane_frame(4, 4, z0, q=0.7)
    t = orc.integrate([inv], [sil], [db], [q], (G, G, G), LIMIT)
    zc = (np.float32(31) + np.float32(0.5)) * np.float32(1.0 / G)
    s = np.float32(zc - np.float32(z0))
    w = np.float32(0.7)
    exp = np.float32(np.float32(np.float32(LIMIT) * np.float32(0) + w * s) / np.float32(np.float32(0) + w))
    assert t[31, 0, 0] == exp and abs(float(exp) + 0.004) < 1e-6
    assert t[0, 0, 0] == np.float32(-LIMIT) and t[G - 1, 0, 0] == np.float32(LIMIT)


def test_two_sensors_weighted_mean_and_order(orc):
    G = 64
    inv = identity_inverse(G)
    za = (31 + 0.5) / G + 0.003
    zb = (31 + 0.5) / G - 0.002
    fa, fb = plane_frame(4, 4, za, q=1.0), plane_frame(4, 4, zb, q=3.0)
    run = lambda fr: orc.integrate([inv, inv], [f[0] for f in fr], [f[1] for f in fr], [f[2] for f in fr],
                                   (G, G, G), LIMIT)
    t_ab, t_ba = run([fa, fb]), run([fb, fa])
    zc = (np.float32(31) + np.float32(0.5)) * np.float32(1.0 / G)
    sa, sb = float(zc - np.float32(za)), float(zc - np.float32(zb))
    assert abs(t_ab[31, 1, 1] - (1.0 * sa + 3.0 * sb) / 4.0) < 1e-7
    assert abs(t_ba[31, 1, 1] - t_ab[31, 1, 1]) < 1e-7      # in-band mean commutes up to rounding
    # order dependence through the overwrite branch: a voxel in front of A's surface
    # but in B's band keeps B's value only if B comes last
    zf = (40 + 0.5) / G
    far_, near_ = plane_frame(4, 4, zf + 0.5, q=1.0), plane_frame(4, 4, zf + 0.001, q=1.0)
    t1 = run([far_, near_])      # far: sdist <= -limit -> -limit, then near in band: (-l*0 + s)/1
    t2 = run([near_, far_])      # near in band, then far overwrites with -limit
    assert abs(t1[40, 0, 0] + 0.001) < 1e-6
    assert t2[40, 0, 0] == np.float32(-LIMIT)


def test_silhouette_overwrites_only_an_untouched_voxel(orc):
    # tsdf_integration.vs:33-38: `if (silhouette < 1.0) { if (weighted_tsd >= limit) { weighted_tsd = -limit; continue; } }`
    # -- a sensor whose silhouette is 0 there clears the voxel only while no sensor has written it; once a value
    # is in the band the same sensor is evaluated like any other (its depth and quality are still sampled)
    G = 64
    inv = identity_inverse(G)
    zc = (31 + 0.5) / G
    seen = plane_frame(4, 4, zc - 0.004, q=1.0)              # sdist +0.004, silhouette 1
    sil0 = list(plane_frame(4, 4, zc - 0.008, q=1.0))        # sdist +0.008 but silhouette 0
    sil0[0] = np.zeros((4, 4), np.float32)
    run = lambda fr: orc.integrate([inv, inv], [f[0] for f in fr], [f[1] for f in fr], [f[2] for f in fr],
                                   (G, G, G), LIMIT)
    assert abs(run([seen, sil0])[31, 1, 1] - 0.006) < 1e-6   # already written: the mean of both
    assert abs(run([sil0, seen])[31, 1, 1] - 0.004) < 1e-6   # untouched: cleared to -limit (weight 0), then `seen` alone
    only = orc.integrate([inv], [sil0[0]], [sil0[1]], [sil0[2]], (G, G, G), LIMIT)
    assert only[31, 1, 1] == np.float32(-LIMIT)


def test_sdist_equal_to_the_limit_leaves_the_voxel_untouched(orc):
    # :46-54: `sdist <= -limit` -> -limit, `sdist >= limit` -> nothing (neither value nor weight), else weighted mean.
    # limit = 1/64 and a surface exactly 1/64 in front of / behind the voxel centre: every quantity is exact.
    G, lim = 16, 1.0 / 64
    inv = identity_inverse(G)
    zc = (7 + 0.5) / G
    at_plus, inband = plane_frame(4, 4, zc - lim, q=1.0), plane_frame(4, 4, zc - lim / 2, q=1.0)
    run = lambda fr: orc.integrate([inv, inv], [f[0] for f in fr], [f[1] for f in fr], [f[2] for f in fr],
                                   (G, G, G), lim)
    # sdist == +limit contributes no weight: the in-band sensor's value stands alone, in either order
    assert run([at_plus, inband])[7, 0, 0] == np.float32(lim / 2)
    assert run([inband, at_plus])[7, 0, 0] == np.float32(lim / 2)
    # sdist == -limit overwrites with -limit
    at_minus = plane_frame(4, 4, zc + lim, q=1.0)
    assert run([inband, at_minus])[7, 0, 0] == np.float32(-lim)
    assert run([at_minus, inband])[7, 0, 0] == np.float32(lim / 2)       # (-limit * 0 + 1 * s) / 1


def test_all_invalid_lut_gives_minus_limit(orc):
    G = 8
    inv = np.full((G, G, G, 4), -1.0, np.float32)
    sil = np.ones((4, 4), np.float32)
    sil[0, 0] = 0.0                       # texel (0,0) is what the clamped fetch hits
    db = np.zeros((4, 4, 2), np.float32)
    q = np.ones((4, 4), np.float32)
    t = orc.integrate([inv], [sil], [db], [q], (G, G, G), LIMIT)
    assert np.all(t == np.float32(-LIMIT))


def test_zero_weight_gives_nan(orc):
    # first in-band sample with weight 0: (limit*0 + 0*s)/(0+0) = NaN (SURVEY "hard parts")
    G = 16
    inv = identity_inverse(G)
    sil, db, q = plane_frame(4, 4, (7 + 0.5) / G, q=0.0)
    t = orc.integrate([inv], [sil], [db], [q], (G, G, G), LIMIT)
    assert np.isnan(t[7, 0, 0])


def test_bricks_mask_restricts_integration(orc):
    G, bv = 16, 8
    inv = identity_inverse(G)
    sil, db, q = plane_frame(4, 4, 0.5)
    full = orc.integrate([inv], [sil], [db], [q], (G, G, G), LIMIT)
    mask = np.zeros(8, np.uint8)
    mask[3] = 1                           # brick (1,1,0)
    part = orc.integrate([inv], [sil], [db], [q], (G, G, G), LIMIT, mask, res_bricks=(2, 2, 2),
                         bbox=((0.0, 0.0, 0.0), (1.0, 1.0, 1.0)), brick_size=0.5)
    assert np.array_equal(part[0:8, 8:16, 8:16], full[0:8, 8:16, 8:16])
    part[0:8, 8:16, 8:16] = -LIMIT
    assert np.all(part == np.float32(-LIMIT))


# ---- geometry ---------------------------------------------------------------
def test_geometry_reference_defaults(orc):
    # default operating point (kinect_client.cpp:87-93,208-209): voxel 0.01 over
    # (-1,0,-1)-(1,2.2,1); float division makes the y resolution 221, not 220
    assert orc.volume_res((-1, 0, -1), (1, 2.2, 1), 0.01) == (200, 221, 200)
    assert orc.volume_res((-1, 0, -1), (1, 2, 1), 2.0 / 64) == (64, 64, 64)
    assert abs(orc.adjust_brick_size(0.1, 0.01) - 0.1) < 1e-7
    assert orc.adjust_brick_size(0.104, 0.01) == np.float32(0.01) * np.float32(10.0)
    rb = orc.divide_box((-1, 0, -1), (1, 2.2, 1), orc.adjust_brick_size(0.1, 0.01))
    assert rb[0] in (20, 21) and rb[1] in (22, 23) and rb[2] in (20, 21)
    assert orc.divide_box((-1, 0, -1), (1, 2, 1), 0.25) == (8, 8, 8)


def test_update_occupied_threshold(orc):
    c = np.array([0, 9, 10, 11, 3, 100], np.uint32)
    ids, ratio = orc.update_occupied(c, 10)      # ">= min_voxels" (recon_integration.cpp:437)
    assert list(ids) == [2, 3, 5] and abs(ratio - 0.5) < 1e-7


def test_camera_position_recovers_pinhole(orc, pkg):
    s = pkg.synth.Sensor(1, 4, 64, 53)
    xyz, _ = pkg.synth.forward_luts(s, (16, 13, 16))
    np.testing.assert_allclose(orc.camera_pos(xyz), s.pos, atol=2e-4)


# ---- hole filling of the ray-marched frame (fillColors) ----------------------
def test_fill_layout_matches_viewlod(orc):
    n, fw, off, res = orc.fill_layout(1280, 720)
    assert n == 10 and fw == 1920                       # 1 + floor(log2(720)), 1.5 * W
    assert tuple(res[0]) == (1280, 720) and tuple(res[1]) == (640, 360) and tuple(res[9]) == (2, 1)
    assert tuple(off[0]) == (0, 0) and tuple(off[1]) == (1280, 360) and tuple(off[2]) == (1280, 180)
    assert np.all(off[n:] == 0) and np.all(res[n:] == 0)


def test_fill_colors_keeps_covered_pixels_and_fills_holes(orc):
    W, H = 64, 48
    rng = np.random.default_rng(4)
    color = np.zeros((H, W, 4), np.float32)
    color[..., :3] = rng.random((H, W, 3))
    color[..., 3] = 1.0
    depth = np.full((H, W), 0.5, np.float32)
    # a hole in the cleared state the ray-marcher leaves: (0,1,0,0), depth 1
    color[20:24, 30:34] = (0.0, 1.0, 0.0, 0.0)
    depth[20:24, 30:34] = 1.0
    oc, od = orc.fill_colors(color, depth)
    covered = np.ones((H, W), bool)
    covered[20:24, 30:34] = False
    assert np.array_equal(oc[covered], color[covered])          # level 0 hit: unchanged
    assert np.array_equal(od, depth)                            # depth comes from LOD 0 always
    hole = oc[20:24, 30:34]
    assert np.all(hole[..., 3] > 0) and np.all(hole[..., :3] >= 0) and np.all(hole[..., :3] <= 1)
    assert not np.any((hole[..., 0] == 0) & (hole[..., 1] == 1) & (hole[..., 2] == 0))   # no clear colour left


def test_fill_colors_prefers_far_samples(orc):
    # tsdf_inpaint.fs keeps samples with depth >= the mean depth of the valid ones: the hole
    # between a near red and a far blue surface is filled with blue
    W, H = 32, 32
    rng = np.random.default_rng(8)
    jitter = (rng.random((H, W)) * 1e-3).astype(np.float32)    # exactly equal depths make sum/n round above
    color = np.zeros((H, W, 4), np.float32)                    # every sample and the shader divides 0 by 0
    depth = np.ones((H, W), np.float32)
    color[:, :14] = (1, 0, 0, 1)
    depth[:, :14] = 0.2 + jitter[:, :14]
    color[:, 18:] = (0, 0, 1, 1)
    depth[:, 18:] = 0.8 + jitter[:, 18:]
    color[:, 14:18] = (0, 1, 0, 0)
    oc, _ = orc.fill_colors(color, depth)
    mid = oc[8:24, 15:17, :3]
    assert np.all(np.isfinite(mid)) and np.all(mid[..., 2] > mid[..., 0])


def test_inpaint_marks_colourless_surface_and_keeps_the_clear_colour(orc):
    """tsdf_inpaint.fs:60-69: a 4 x 4 neighbourhood without a single alpha > 0 sample passes the depth on and writes
    (0, 0, 0, -1) over a surface (depth < 1: the ray-marcher's fallback blend has alpha -1) and the clear colour
    (0, 1, 0, 0) over background; LOD 1 of the 1.5 W x H atlas sits at (W, H - H/2) (ViewLod::setResolution)"""
    W, H = 32, 16
    color = np.zeros((H, W, 4), np.float32)
    depth = np.ones((H, W), np.float32)
    color[:, :16] = (0.3, 0.4, 0.5, -1.0)            # left half: a surface nobody coloured
    depth[:, :16] = 0.5
    color[:, 16:] = (0.0, 1.0, 0.0, 0.0)             # right half: cleared
    _, od, atlas = orc.fill_colors(color, depth, return_atlas=True)
    assert atlas.shape == (H, 48, 4) and np.array_equal(od, depth)
    lod1 = atlas[H - H // 2:H, W:W + W // 2]          # 16 x 8 texels
    assert np.all(lod1[2:6, 1:6] == np.array([0.0, 0.0, 0.0, -1.0], np.float32))
    assert np.all(lod1[2:6, 10:15] == np.array([0.0, 1.0, 0.0, 0.0], np.float32))


def test_oracle_is_clean_under_asan_and_ubsan():
    """the C restatement driven over every entry point (NaN / inf / invalid-LUT inputs
    included) under AddressSanitizer + UBSan; GPU sanitizers do not exist on the pool"""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["make", "-s", "-C", os.path.join(root, "oracle"), "asan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "selftest done" in r.stdout and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
