"""Known-answer tests of the oracle's pre_* passes, derived by hand from the shader text (the reference has
no tests or vectors for them and the GLSL cannot run here).  Every expected value below is
worked out from the GLSL source cited next to it, not from the oracle.  tests/mutation_check.py flips each of
the reference's quirks in a scratch copy of the oracle and shows that at least one of these tests goes red.

  pre_depth.fs    bilateral filter (:85-127), u8 uncompress (:51-61)
  inc_bricks.glsl mark_brick (:40-58)
  pre_normal.fs   calculate_normal (:26-56)
  pre_quality.fs  bilateral_filter (:64-118), normal_angle (:43-48)
"""
import math

import numpy as np

F = np.float32
BMIN, BMAX = (-1.0, 0.0, -1.0), (1.0, 2.0, 1.0)
LIMITS = (0.5, 4.5)


def const_lut(pos, ch=3, res=(2, 2, 2)):
    """calibration volume that returns `pos` for every lookup"""
    return np.tile(np.asarray(pos, np.float32), (res[2], res[1], res[0], 1)).reshape(res[2], res[1], res[0], ch)


def identity_lut(n=8):
    """cv_xyz with world = (u, v, d) at the texel centres: LINEAR filtering reproduces the identity for
    coordinates inside [0.5/n, 1 - 0.5/n] (GL 4.4 section 8.14)"""
    c = (np.arange(n) + 0.5) / n
    D, V, U = np.meshgrid(c, c, c, indexing="ij")
    return np.stack([U, V, D], axis=-1).astype(np.float32)


def run_pre_depth(orc, depth, filter_textures=True, compress=False, near=0.5, far=4.5):
    h, w = depth.shape
    color = np.full((h, w, 3), 128, np.uint8)
    rg, _ = orc.pre_depth(depth, color, const_lut((0.0, 1.0, 0.0)), const_lut((0.5, 0.5), ch=2), LIMITS, BMIN, BMAX,
                          filter_textures, compress, near, far)
    return rg


# ---- pre_depth.fs: bilateral filter -----------------------------------------------------------------
def test_bilateral_constant_plane_is_a_fixed_point(orc):
    # every tap equals the centre: gauss_range = 1 (:45-48), so depth_bf = d * sum(w_s) and, for d = 2 (a power of
    # two: every product and partial sum is exactly doubled), filtered = 2 exactly; normalised (2 - 0.5) / 4.
    # w_range = 169 taps of 1.0 -> quality channel 169 / 169 = 1 (:126).  Border pixels clamp their taps
    # (CLAMP_TO_EDGE) to the same value.
    rg = run_pre_depth(orc, np.full((20, 20), 2.0, np.float32))
    assert np.all(rg[..., 0] == F(0.375)) and np.all(rg[..., 1] == F(1.0))


def test_bilateral_rejects_taps_across_a_depth_step(orc):
    # dist_range_max = 0.35 * depth / 4.5 (:89-92): 0.156 m at 2 m, 0.311 m at 4 m -- a 2 m step is rejected from
    # both sides (:106), the accepted taps all equal the centre, so the depth is unchanged and the quality
    # channel is (#accepted) / 169: num_samples counts rejected taps too (:100,126)
    d = np.full((20, 24), 2.0, np.float32)
    E = 12
    d[:, E:] = 4.0
    rg = run_pre_depth(orc, d)
    for col, depth_norm in ((E - 1, 0.375), (E, 0.875)):
        assert rg[10, col, 0] == F(depth_norm)
        assert rg[10, col, 1] == F(91.0) / F(169.0)            # 7 of 13 columns
    assert rg[10, E - 4, 1] == F(130.0) / F(169.0)              # 10 of 13 columns
    assert rg[10, E - 7, 1] == F(1.0) and rg[10, E + 6, 1] == F(1.0)
    assert rg[10, E - 7, 0] == F(0.375) and rg[10, E + 6, 0] == F(0.875)


def test_bilateral_range_threshold_scales_with_depth(orc):
    # a tap 0.2 m behind the centre: inside the threshold at 4 m (0.311), outside at 2 m (0.156)
    for centre, inside in ((4.0, True), (2.0, False)):
        d = np.full((13, 13), centre, np.float32)
        d[6, 7] = centre + 0.2
        rg = run_pre_depth(orc, d)
        if inside:
            gr = 1.0 - 0.2 / (0.35 * centre / 4.5)
            assert abs(rg[6, 6, 1] - (168 + gr) / 169) < 2e-6
        else:
            assert rg[6, 6, 1] == F(168.0) / F(169.0)


def test_bilateral_spatial_weights_go_negative_in_the_corners(orc):
    # computeGaussSpace = 1 - length(x, y) / 6 (:37-41) is NOT clamped: the four corners (+-6, +-6) weigh
    # 1 - sqrt(72) / 6 = -0.414.  Only the centre and the corners are valid (all other taps are 0 m: outside
    # [cv_min_ds, cv_max_ds], :72-74,106); the corners lie 0.1 m behind the centre.
    d = np.zeros((13, 13), np.float32)
    c, delta = 4.0, 0.1
    d[6, 6] = c
    for y, x in ((0, 0), (0, 12), (12, 0), (12, 12)):
        d[y, x] = c + delta
    rg = run_pre_depth(orc, d)
    drm = 0.35 * c / 4.5
    gr = 1.0 - delta / drm
    gs = 1.0 - math.sqrt(72.0) / 6.0
    w = 1.0 + 4 * gs * gr
    filtered = (c + 4 * gs * gr * (c + delta)) / w
    assert gs < 0 and w < 0 and filtered > c + delta  # not a convex combination: the result leaves [c, c + delta]
    assert abs(rg[6, 6, 0] - (filtered - 0.5) / 4.0) < 1e-5
    assert abs(rg[6, 6, 1] - (1 + 4 * gr) / 169) < 1e-6


def test_bilateral_skips_taps_outside_the_calibrated_range(orc):
    # is_outside(depth_s) uses cv_min_ds / cv_max_ds (:72-74): 0.3 m and 5 m taps never contribute
    d = np.full((13, 13), 0.6, np.float32)                # threshold at 0.6 m: 0.0467
    d[6, 0:3] = 0.49                                       # < cv_min_ds (and also beyond the range threshold)
    d[0, 0] = 0.62                                         # valid
    rg = run_pre_depth(orc, d)
    gr = 1.0 - 0.02 / (0.35 * 0.6 / 4.5)
    assert abs(rg[6, 6, 1] - (165 + gr) / 169) < 2e-6
    # a tap that ONLY the calibrated range excludes: centre 0.52 m (threshold 0.0404), taps at 0.49 m are 0.03 away
    d = np.full((13, 13), 0.52, np.float32)
    d[3, 2:6] = 0.49
    rg = run_pre_depth(orc, d)
    assert rg[6, 6, 1] == F(165.0) / F(169.0) and abs(rg[6, 6, 0] - 0.02 / 4.0) < 1e-6
    # ... and likewise above cv_max_ds: centre 4.45 m (threshold 0.346), taps at 4.6 m
    d = np.full((13, 13), 4.45, np.float32)
    d[9, 2:7] = 4.6
    assert run_pre_depth(orc, d)[6, 6, 1] == F(164.0) / F(169.0)


def test_unfiltered_and_out_of_box_outputs(orc):
    # !filter_textures -> (depth_norm, 1) (:149-151); a world position outside the box -> (0, 0) (:144-147)
    d = np.full((4, 4), 2.5, np.float32)
    assert np.all(run_pre_depth(orc, d, filter_textures=False) == np.array([0.5, 1.0], np.float32))
    color = np.zeros((4, 4, 3), np.uint8)
    rg, _ = orc.pre_depth(d, color, const_lut((0.0, 2.5, 0.0)), const_lut((0.5, 0.5), ch=2), LIMITS, BMIN, BMAX, True)
    assert np.all(rg == 0.0)
    rg, _ = orc.pre_depth(d, color, const_lut((1.0, 2.0, -1.0)), const_lut((0.5, 0.5), ch=2), LIMITS, BMIN, BMAX, False)
    assert np.all(rg == np.array([0.5, 1.0], np.float32))          # the box is closed: <= and >= (inc_bbox_test.glsl)


# ---- pre_depth.fs: u8 uncompress ---------------------------------------------------------------------
def test_uncompress_thresholds_and_sqrt_mapping(orc):
    # uncompress (:51-61): d_c < scaled_near -> 0, else (d_c^2 + 0.15 * scaled_near) * scale + near, with
    # scale = far - near, scaled_near = scale / 255 (NetKinectArray.cpp:346-351).  near 0.5, far 4.5.
    codes = np.array([[0, 3, 4, 128, 255]], np.float32) / F(255.0)
    rg = run_pre_depth(orc, codes, filter_textures=False, compress=True)
    scale, sn = 4.0, 4.0 / 255.0
    below = (0.0 - 0.5) / 4.0                                  # uncompressed 0 m, normalised
    assert rg[0, 0, 0] == F(below) and rg[0, 1, 0] == F(below)    # 0 and 3/255 < 4/255
    for i, code in ((2, 4), (3, 128), (4, 255)):                   # 4/255 == scaled_near is NOT below the threshold
        dc = code / 255.0
        metres = (dc * dc + 0.15 * sn) * scale + 0.5
        assert abs(rg[0, i, 0] - (metres - 0.5) / 4.0) < 1e-6
    assert abs(rg[0, 4, 0] - (1.0 + 0.15 * sn)) < 1e-6          # code 255: 4.5 m + 0.15 * 4/255 * 4 m


# ---- inc_bricks.glsl: mark_brick ----------------------------------------------------------------------
BS = 0.25                      # brick edge: 8 x 8 x 8 bricks over the 2 m box
RB = (8, 8, 8)


def mark(orc, pos):
    """brick counters after mark_brick(pos): one pixel whose world position is `pos`"""
    counters = np.zeros(RB[0] * RB[1] * RB[2], np.uint32)
    db = np.zeros((1, 1, 2), np.float32)
    db[0, 0, 0] = 0.5
    orc.normal(db, const_lut(pos), BMIN, BMAX, BS, RB, counters)
    return {tuple(int(v) for v in np.unravel_index(i, (RB[2], RB[1], RB[0]))[::-1]): int(counters[i])
            for i in np.nonzero(counters)[0]}


def centre(ix, iy, iz):
    # to_world(vec3(0.5), index) = index * brick_size + bbox_min + 0.5 * brick_size (:27-29)
    return np.array([ix * BS + BMIN[0] + 0.5 * BS, iy * BS + BMIN[1] + 0.5 * BS, iz * BS + BMIN[2] + 0.5 * BS])


def test_mark_brick_centre_counts_once(orc):
    # difference = 0: min_c = (1,1,1) (0 < 0 is false), sign(0) = 0 -> the "neighbour" is the brick itself and
    # its increment is (d_abs.x > 0.1 * brick_size) = 0 (:53); the home brick gets 1 (:58)
    assert mark(orc, centre(2, 3, 4)) == {(2, 3, 4): 1}


def test_mark_brick_neighbour_across_the_dominant_axis(orc):
    # x dominant and |dx| = 0.1 > 0.025: neighbour (3,3,4) += 1, home += 1
    assert mark(orc, centre(2, 3, 4) + (0.1, 0.01, -0.02)) == {(2, 3, 4): 1, (3, 3, 4): 1}
    assert mark(orc, centre(2, 3, 4) + (-0.1, 0.01, -0.02)) == {(2, 3, 4): 1, (1, 3, 4): 1}
    # z dominant, |dx| = 0.03 > 0.025: the neighbour across z is incremented
    assert mark(orc, centre(2, 3, 4) + (0.03, 0.0, 0.12)) == {(2, 3, 4): 1, (2, 3, 5): 1}


def test_mark_brick_increment_tests_d_abs_x_whatever_the_dominant_axis(orc):
    # the reference's quirk (:53): the neighbour's increment is (d_abs.x > brick_size * 0.1) even when the
    # neighbour lies across y or z -- with |dx| = 0.01 nothing is added although |dy| = 0.11
    assert mark(orc, centre(2, 3, 4) + (0.01, -0.11, 0.02)) == {(2, 3, 4): 1}
    assert mark(orc, centre(2, 3, 4) + (0.01, 0.02, 0.11)) == {(2, 3, 4): 1}
    # ... and with |dx| = 0.026 it is
    assert mark(orc, centre(2, 3, 4) + (0.026, -0.11, 0.02)) == {(2, 3, 4): 1, (2, 2, 4): 1}


def test_mark_brick_ties_select_every_maximal_axis(orc):
    # d_abs.x == d_abs.y == max: min_c = (1,1,0) -> diagonal neighbour (:47-52); 3/32 is exact in binary32
    assert mark(orc, centre(2, 3, 4) + (0.09375, 0.09375, 0.0)) == {(2, 3, 4): 1, (3, 4, 4): 1}
    assert mark(orc, centre(2, 3, 4) + (0.09375, -0.09375, -0.09375)) == {(2, 3, 4): 1, (3, 2, 3): 1}


def test_mark_brick_neighbour_is_clamped_to_the_grid(orc):
    # clamp(index + offset, 0, resolution - 1) (:53): at the last brick of x the neighbour is the brick itself
    assert mark(orc, centre(7, 3, 4) + (0.1, 0.0, 0.0)) == {(7, 3, 4): 2}
    assert mark(orc, centre(0, 0, 0) + (0.0, -0.1, 0.0)) == {(0, 0, 0): 1}       # |dx| = 0: clamped AND no increment
    assert mark(orc, centre(0, 0, 0) + (-0.1, 0.0, 0.0)) == {(0, 0, 0): 2}


def test_mark_brick_outside_the_grid_is_skipped(orc):
    # the reference converts a negative float to uvec3 (undefined) and indexes out of range; this build's
    # stated decision: such positions mark nothing
    assert mark(orc, (-1.2, 1.0, 0.0)) == {}
    assert mark(orc, (0.0, 2.1, 0.0)) == {}


# ---- pre_normal.fs -------------------------------------------------------------------------------------
def test_normal_of_a_tilted_plane_points_at_the_sensor(orc):
    # world = (u, v, d) (identity LUT), depth d = 0.4 + a * px.  world_b - world_t = (0, -2/H, 0),
    # world_l - world_r = (-2/W, 0, -2a); cross(b - t, l - r) = (4a/H, 0, -4/(H W)) (:55) -> normalize((a W, 0, -1)):
    # the normal points to -z, towards the sensor (smaller depth), and leans to +x where depth grows with x.
    W = H = 8
    a = 0.01
    db = np.zeros((H, W, 2), np.float32)
    db[..., 0] = 0.4 + a * np.arange(W)[None, :]
    n = orc.normal(db, identity_lut(8), (-9, -9, -9), (9, 9, 9), 1.0, (1, 1, 1))
    want = np.array([a * W, 0.0, -1.0])
    want /= np.linalg.norm(want)
    np.testing.assert_allclose(n[4, 4], want, atol=2e-5)
    np.testing.assert_allclose(n[2, 5], want, atol=2e-5)
    assert n[4, 4, 2] < -0.99 and n[4, 4, 0] > 0.07


def test_normal_replaces_invalid_neighbours_by_the_centre_depth(orc):
    # is_outside(depth_r) ? depth : depth_r (:43-46): with the right neighbour invalid the x difference spans one
    # pixel of depth instead of two: l - r = (-2/W, 0, -a) -> normalize((a W / 2, 0, -1))
    W = H = 8
    a = 0.01
    db = np.zeros((H, W, 2), np.float32)
    db[..., 0] = 0.4 + a * np.arange(W)[None, :]
    db[4, 5, 0] = 0.0                                   # right neighbour of (4, 4)
    n = orc.normal(db, identity_lut(8), (-9, -9, -9), (9, 9, 9), 1.0, (1, 1, 1))
    want = np.array([a * W / 2, 0.0, -1.0])
    want /= np.linalg.norm(want)
    np.testing.assert_allclose(n[4, 4], want, atol=2e-5)
    assert np.all(n[4, 5] == 0.0)                       # an invalid pixel itself: vec3(0) (:28-30)
    db[4, 5, 0] = 1.0                                   # d >= 1 is outside too (:22-24)
    assert np.all(orc.normal(db, identity_lut(8), (-9, -9, -9), (9, 9, 9), 1.0, (1, 1, 1))[4, 5] == 0.0)


# ---- pre_quality.fs ------------------------------------------------------------------------------------
def quality_at(orc, db, normal, cam, px=8, py=8):
    h, w = db.shape[:2]
    nrm = np.tile(np.asarray(normal, np.float32), (h, w, 1))
    return float(orc.quality(db, nrm, identity_lut(8), cam)[py, px])


def frame(d, size=17):
    db = np.zeros((size, size, 2), np.float32)
    db[..., 0] = d
    return db


def cam_in_front(d, px=8, py=8, size=17):
    # camera on the pixel's own ray, 2 units in front: normalize(cam - world) = (0, 0, -1)
    return ((px + 0.5) / size, (py + 0.5) / size, d - 2.0)


def test_quality_of_a_fronto_parallel_plane(orc):
    # no border taps, every gauss_range = 1: lateral_quality = 1, (w_range / 169)^6 = 1, so
    # quality = 1 / (depth * 6.5) * angle^2 (:107-112) with angle = dot(normalize(cam - world), normal) (:43-48)
    for d in (0.25, 0.5):
        assert abs(quality_at(orc, frame(d), (0, 0, -1), cam_in_front(d)) - 1.0 / (6.5 * d)) < 1e-6
    # a normal 60 degrees off the viewing ray: angle = 0.5 -> a quarter
    tilted = (math.sin(math.pi / 3), 0.0, -math.cos(math.pi / 3))
    assert abs(quality_at(orc, frame(0.5), tilted, cam_in_front(0.5)) - 0.25 / 3.25) < 1e-6
    # a back-facing normal: pow(angle, 2) keeps the magnitude
    assert abs(quality_at(orc, frame(0.5), (0, 0, 1), cam_in_front(0.5)) - 1.0 / 3.25) < 1e-6


def test_quality_counts_invalid_and_distant_taps_as_border(orc):
    # is_outside(depth_s) || depth_range > 0.35 * depth (:96): border_samples; lateral = 1 - k/169,
    # w_range counts the others: quality = (1 - k/169)^6 * ((169 - k)/169)^6 / (6.5 d)
    d = 0.5
    db = frame(d)
    db[2:15, 3, 0] = 0.0                                   # one column of the window invalid: k = 13
    want = (156 / 169) ** 12 / (6.5 * d)
    assert abs(quality_at(orc, db, (0, 0, -1), cam_in_front(d)) - want) < 2e-6
    db = frame(d)
    db[2:15, 3, 0] = d + 0.36 * d                          # beyond the range threshold: border as well
    assert abs(quality_at(orc, db, (0, 0, -1), cam_in_front(d)) - want) < 2e-6
    db[2:15, 3, 0] = 1.0                                   # depth 1.0 is outside (d >= 1)
    assert abs(quality_at(orc, db, (0, 0, -1), cam_in_front(d)) - want) < 2e-6
    # taps that ONLY is_outside excludes: centre 0.9 (threshold 0.315), a column at exactly 1.0 / at 1.05
    for outside in (1.0, 1.05):
        db = frame(0.9)
        db[2:15, 3, 0] = outside
        want9 = (156 / 169) ** 12 / (6.5 * 0.9)
        assert abs(quality_at(orc, db, (0, 0, -1), cam_in_front(0.9)) - want9) < 2e-6


def test_quality_range_weight_of_near_taps(orc):
    # a tap at depth d + 0.1 d: gauss_range = 1 - 0.1 d / (0.35 d) = 0.7143 (:33-36,74-75)
    d = 0.5
    db = frame(d)
    db[8, 9, 0] = d * 1.1
    gr = 1.0 - 0.1 / 0.35
    want = ((168 + gr) / 169) ** 6 / (6.5 * d)
    assert abs(quality_at(orc, db, (0, 0, -1), cam_in_front(d)) - want) < 2e-6


def test_quality_is_zero_for_invalid_pixels(orc):
    for bad in (0.0, -0.2, 1.0, 1.5):                      # is_outside: d <= 0 || d >= 1 (:39-41,67-69)
        db = frame(0.5)
        db[8, 8, 0] = bad
        assert quality_at(orc, db, (0, 0, -1), cam_in_front(0.5)) == 0.0
