"""Parity of the HIP path (through the C ABI) with the CPU oracle on the same
seeded inputs.  All comparisons are bit-exact on the float values (NaN == NaN,
-0 == +0): both sides evaluate the same IEEE binary32 operations in the same
order, with FMA contraction off."""
import ctypes as C

import numpy as np
import pytest

from conftest import count_diff, same_bits

pytestmark = pytest.mark.gpu

BMIN, BMAX = (-1.0, 0.0, -1.0), (1.0, 2.0, 1.0)
IMG = {"morph": 1, "depth_rg": 2, "lab": 3, "depth_b": 4, "sil": 5, "normal": 6, "quality": 7}


def build(pkg, n=2, wh=(128, 106), G=64, inv_res=None, lut_res=(32, 27, 32), seed=1234, **cfgkw):
    capi, synth = pkg.capi, pkg.synth
    scene = synth.Scene(n, wh[0], wh[1], lut_res=lut_res, seed=seed)
    kw = dict(voxel_size=2.0 / G, brick_size=8 * 2.0 / G)
    kw.update(cfgkw)
    cfg = capi.make_config(n, wh, **kw)
    ctx = capi.Context(cfg, 0)
    g = ctx.geo
    inv_res = inv_res or tuple(g.res_volume)
    inv = scene.inverse(inv_res)
    for i in range(n):
        ctx.set_calibration(i, scene.xyz[i], lut_res, scene.uv[i], lut_res, (0.5, 4.5))
        ctx.set_inverse_calibration(i, inv[i], inv_res)
    return scene, ctx, inv


def oracle_run(orc, scene, ctx, inv, **kw):
    g = ctx.geo
    f = ctx.cfg.flags
    args = dict(limit=ctx.cfg.tsdf_limit, brick_size=g.brick_size, bv=g.brick_voxels, res_bricks=tuple(g.res_bricks),
                min_voxels=ctx.cfg.min_voxels_per_brick, filter_textures=bool(f & 1), processed=bool(f & 2),
                refine=bool(f & 4), use_bricks=bool(f & 8))
    args.update(kw)
    return orc.run_pipeline(scene, BMIN, BMAX, tuple(g.res_volume), inv, **args)


def check_images(ctx, ref, n):
    for name, which in IMG.items():
        for i in range(n):
            got = ctx.readback_image(which, i)
            assert same_bits(got, ref[name][i]), "%s sensor %d: %d texels differ" % (
                name, i, count_diff(got, ref[name][i]))


# ---------------------------------------------------------------------------
@pytest.mark.parametrize("flags", [15, 14, 13, 11, 8, 7])
def test_every_pass_and_tsdf_bit_exact(pkg, orc, flags):
    scene, ctx, inv = build(pkg, flags=flags)
    ctx.step(scene.depth, scene.color)
    ref = oracle_run(orc, scene, ctx, inv)
    check_images(ctx, ref, 2)
    assert np.array_equal(ctx.readback_brick_counters(), ref["counters"])
    ids, ratio = ctx.get_occupied()
    assert np.array_equal(ids, ref["occupied"]) and ratio == np.float32(ref["ratio"])
    assert abs(ctx.occupied_ratio() - ref["ratio"]) < 1e-7
    got = ctx.readback_tsdf()
    assert same_bits(got, ref["tsdf"]), "%d voxels differ" % count_diff(got, ref["tsdf"])
    assert np.any(np.abs(got) < 0.01) and np.any(got == np.float32(-0.01))
    ctx.close()


def test_compressed_u8_depth(pkg, orc):
    capi, synth = pkg.capi, pkg.synth
    scene, ctx, inv = build(pkg, compress_depth=1, flags=13)   # u8 + morph is incoherent in the reference (A.5)
    d8 = synth.compress_depth_u8(scene.depth)
    ctx.step(d8, scene.color)
    unit = (d8.astype(np.float32) / np.float32(255.0)).astype(np.float32)
    ref = oracle_run(orc, scene, ctx, inv, compress=True, depth_override=unit)
    check_images(ctx, ref, 2)
    assert same_bits(ctx.readback_tsdf(), ref["tsdf"])
    assert np.any(np.abs(ref["tsdf"]) < 0.01)
    ctx.close()


@pytest.mark.parametrize("n", [1, 3, 5, 8])
def test_sensor_counts(pkg, orc, n):
    scene, ctx, inv = build(pkg, n=n, wh=(64, 53), G=32, lut_res=(16, 13, 16))
    ctx.step(scene.depth, scene.color)
    ref = oracle_run(orc, scene, ctx, inv)
    check_images(ctx, ref, n)
    assert same_bits(ctx.readback_tsdf(), ref["tsdf"])
    assert np.sum(np.abs(ref["tsdf"]) < 0.01) > 50 and len(ref["occupied"]) > 0      # scene is not degenerate
    ctx.close()


@pytest.mark.parametrize("wh,G", [((5, 3), 16), ((17, 1), 16), ((1, 1), 8), ((13, 13), 24), ((33, 17), 32), ((70, 53), 32), ((16, 16), 9),
                                  ((70, 53), 1), ((70, 53), 2), ((70, 53), 3), ((70, 53), 5), ((70, 53), 7)])      # grids smaller than one tile
def test_images_smaller_than_a_block_or_the_filter_window(pkg, orc, wh, G):
    """sensors of a few pixels (narrower than the 16 x 16 block, than the 13 x 13 window, a single row, a single pixel) and grids
    that are no multiple of the 8-voxel tile: every clamp at an image border and every partial block / tile is exercised; both
    sweeps, images and brick table equal the oracle's"""
    scene, ctx, inv = build(pkg, n=2, wh=wh, G=G, lut_res=(8, 7, 8), tsdf_limit=0.15, min_voxels=1)   # (a wide band: a few pixels reach voxels)
    for bricks in (True, False):
        ctx.set_use_bricks(bricks)
        ctx.step(scene.depth, scene.color)
        ref = oracle_run(orc, scene, ctx, inv, use_bricks=bricks)
        check_images(ctx, ref, 2)
        assert np.array_equal(ctx.readback_brick_counters(), ref["counters"])
        assert same_bits(ctx.readback_tsdf(), ref["tsdf"]), (wh, G, bricks, count_diff(ctx.readback_tsdf(), ref["tsdf"]))
    ctx.close()


@pytest.mark.parametrize("lut_res", [(1, 1, 1), (2, 2, 2), (3, 1, 2), (1, 5, 1)])
def test_calibration_volumes_of_one_or_two_cells(pkg, orc, lut_res):
    """cv_xyz / cv_uv volumes with a single cell along some or all axes: every LINEAR lookup clamps both taps to the same
    texel, camera position and frustum planes come from coincident corner samples (NaN and all) -- images, counters and
    volume still equal the oracle's"""
    scene, ctx, inv = build(pkg, wh=(64, 53), G=16, lut_res=lut_res, seed=5)
    assert np.array_equal(ctx.camera_position(0), orc.camera_pos(scene.xyz[0]), equal_nan=True)
    for bricks in (True, False):
        ctx.set_use_bricks(bricks)
        ctx.step(scene.depth, scene.color)
        ref = oracle_run(orc, scene, ctx, inv, use_bricks=bricks)
        check_images(ctx, ref, 2)
        assert np.array_equal(ctx.readback_brick_counters(), ref["counters"])
        assert same_bits(ctx.readback_tsdf(), ref["tsdf"])
    ctx.close()


@pytest.mark.parametrize("resample", [True, False])
@pytest.mark.parametrize("G,inv_res", [(64, (45, 45, 45)), (64, (90, 70, 80)), (50, None), (40, (64, 64, 64))])
def test_generic_inverse_lut_resolution(pkg, orc, G, inv_res, resample):
    """inverse LUT at a resolution != the TSDF grid (and a non-power-of-two grid):
    the 8-tap trilinear lookup, either evaluated once at upload into the grid layout
    (default) or per frame (RGBDR_FLAG_NO_RESAMPLE)"""
    flags = 15 | (0 if resample else pkg.capi.FLAG_NO_RESAMPLE)
    scene, ctx, inv = build(pkg, G=G, inv_res=inv_res, flags=flags)
    for bricks in (True, False):
        ctx.set_use_bricks(bricks)
        ctx.step(scene.depth, scene.color)
        ref = oracle_run(orc, scene, ctx, inv, use_bricks=bricks)
        got = ctx.readback_tsdf()
        assert same_bits(got, ref["tsdf"]), "%d voxels differ" % count_diff(got, ref["tsdf"])
        assert np.sum(np.abs(got) < 0.01) > 100
    ctx.close()


def test_reference_default_grid(pkg, orc):
    """reference operating point: voxel 0.01, brick 0.1 (10 voxels, not a tile
    multiple), bbox y up to 2.2 -> 200 x 221 x 200"""
    capi, synth = pkg.capi, pkg.synth
    scene = synth.Scene(2, 64, 53, lut_res=(16, 13, 16))
    bmax = (1.0, 2.2, 1.0)
    cfg = capi.make_config(2, (64, 53), bbox_max=bmax, voxel_size=0.01, brick_size=0.1)
    ctx = capi.Context(cfg, 0)
    g = ctx.geo
    assert tuple(g.res_volume) == (200, 221, 200) and g.brick_voxels == 10
    inv_res = (70, 77, 70)
    inv = [synth.inverse_lut(s, inv_res, BMIN, bmax) for s in scene.sensors]
    for i in range(2):
        ctx.set_calibration(i, scene.xyz[i], (16, 13, 16), scene.uv[i], (16, 13, 16), (0.5, 4.5))
        ctx.set_inverse_calibration(i, inv[i], inv_res)
    ctx.step(scene.depth, scene.color)
    assert tuple(g.res_bricks) == (20, 22, 20)           # divideBox: 22 bricks on y, not ceil(221 / 10)
    ref = orc.run_pipeline(scene, BMIN, bmax, tuple(g.res_volume), inv, limit=0.01, brick_size=g.brick_size,
                           res_bricks=tuple(g.res_bricks))
    assert np.array_equal(ctx.readback_brick_counters(), ref["counters"])
    ids, ratio = ctx.get_occupied()
    assert np.array_equal(ids, ref["occupied"]) and ratio == np.float32(len(ids)) / np.float32(8800)
    got = ctx.readback_tsdf()
    assert same_bits(got, ref["tsdf"]), "%d voxels differ" % count_diff(got, ref["tsdf"])
    assert np.sum(np.abs(got) < 0.01) > 1000 and ref["counters"].sum() > 100
    ctx.close()


@pytest.mark.parametrize("case", ["default", "default-generic-kernel", "overflow", "overflow-generic-kernel"])
def test_brick_sweep_draws_the_reference_index_lists(pkg, orc, case):
    """Brick mode against the reference's own brick -> voxel lists (divideBox + containedVoxels, run
    literally by the oracle) for random sets of occupied bricks.  `default`: the reference's operating
    point, where most bricks also list the first voxel row of the next brick; `overflow`: a box whose last
    x and y bricks list one index past the axis end, which aliases voxels of the next row / slice
    through z*X*Y + y*X + x."""
    capi, synth = pkg.capi, pkg.synth
    over = case.startswith("overflow")
    bmin = (-1.0, 0.0, -1.0)
    bmax = (1.4, 2.4, 1.0) if over else (1.0, 2.2, 1.0)
    voxel = 0.02 if over else 0.01
    flags = capi.FLAGS_DEFAULT | (capi.FLAG_NO_RESAMPLE if case.endswith("generic-kernel") else 0)
    scene = synth.Scene(2, 64, 53, lut_res=(16, 13, 16))
    ctx = capi.Context(capi.make_config(2, (64, 53), bbox_min=bmin, bbox_max=bmax, voxel_size=voxel, brick_size=0.1,
                                        flags=flags), 0)
    g = ctx.geo
    res = tuple(g.res_volume)
    assert res == ((121, 121, 100) if over else (200, 221, 200))
    inv_res = (61, 57, 49) if over else (70, 77, 70)
    inv = [synth.inverse_lut(s, inv_res, bmin, bmax) for s in scene.sensors]
    for i in range(2):
        ctx.set_calibration(i, scene.xyz[i], (16, 13, 16), scene.uv[i], (16, 13, 16), (0.5, 4.5))
        ctx.set_inverse_calibration(i, inv[i], inv_res)
    ctx.step(scene.depth, scene.color)
    ref = orc.run_pipeline(scene, bmin, bmax, res, None, brick_size=g.brick_size, res_bricks=tuple(g.res_bricks))
    assert np.array_equal(ctx.readback_brick_counters(), ref["counters"])
    ctx.set_use_bricks(False)
    ctx.integrate()
    full = ctx.readback_tsdf()
    ctx.set_use_bricks(True)
    rng = np.random.default_rng(11)
    bv = g.brick_voxels
    differs_from_partition = 0
    for density in (0.03, 0.4):
        occ = (rng.random(g.num_bricks) < density).astype(np.uint8)
        occ[-1] = 1                          # the corner brick: the one whose lists overflow
        ctx.set_occupied_bricks(np.nonzero(occ)[0])
        ctx.integrate()
        got = ctx.readback_tsdf()
        want = orc.integrate(inv, ref["sil"], ref["depth_b"], ref["quality"], res, 0.01, occ, res_bricks=tuple(g.res_bricks),
                             bbox=(bmin, bmax), brick_size=g.brick_size)
        assert same_bits(got, want), "%d voxels differ" % count_diff(got, want)
        # the voxels drawn are the full sweep's values, everything else the clear value
        drawn, _, outside = orc.brick_voxel_mask(bmin, bmax, g.brick_size, res, occ)
        drawn = drawn.astype(bool)
        assert same_bits(got[drawn], full[drawn]) and np.all(got[~drawn] == np.float32(-0.01))
        assert (outside > 0) == over
        # ... and NOT what an integer partition voxel // brick_voxels would draw (round 1's membership)
        rb = g.res_bricks
        part = np.kron(np.pad(occ.reshape(rb[2], rb[1], rb[0]), ((0, 1), (0, 1), (0, 1))), np.ones((bv,) * 3, np.uint8))
        part = part[:res[2], :res[1], :res[0]].astype(bool)
        differs_from_partition += int(np.sum(part != drawn))
    assert differs_from_partition > 1000
    ctx.close()


def test_colour_resolution_differs_from_depth_resolution(pkg, orc):
    """rgb_size != depth_size (e.g. Kinect V2: 1280x1080 colour, 512x424 depth), odd sizes"""
    capi, synth = pkg.capi, pkg.synth
    W, H, Wc, Hc = 131, 107, 203, 97
    scene = synth.Scene(2, W, H, lut_res=(32, 27, 32), color_wh=(Wc, Hc))
    assert scene.color.shape == (2, Hc, Wc, 3)
    ctx = capi.Context(capi.make_config(2, (W, H), color_wh=(Wc, Hc), voxel_size=2.0 / 64, brick_size=8 * 2.0 / 64), 0)
    inv = scene.inverse((64, 64, 64))
    for i in range(2):
        ctx.set_calibration(i, scene.xyz[i], (32, 27, 32), scene.uv[i], (32, 27, 32), (0.5, 4.5))
        ctx.set_inverse_calibration(i, inv[i], (64, 64, 64))
    ctx.step(scene.depth, scene.color)
    ref = oracle_run(orc, scene, ctx, inv)
    check_images(ctx, ref, 2)
    assert np.array_equal(ctx.readback_brick_counters(), ref["counters"])
    assert same_bits(ctx.readback_tsdf(), ref["tsdf"])
    assert np.sum(np.abs(ref["tsdf"]) < 0.01) > 200
    ctx.close()


def test_anisotropic_override_grid(pkg, orc):
    """res_override grid (the weak-scaling benchmark grids): voxel edge differs per
    axis, brick membership follows per axis"""
    capi, synth = pkg.capi, pkg.synth
    scene = synth.Scene(2, 64, 53, lut_res=(16, 13, 16))
    res = (32, 64, 128)
    cfg = capi.make_config(2, (64, 53), voxel_size=2.0 / 32, brick_size=8 * 2.0 / 32, res_override=res)
    ctx = capi.Context(cfg, 0)
    g = ctx.geo
    assert tuple(g.res_volume) == res and tuple(g.brick_voxels_axis) == (8, 16, 32) and tuple(g.res_bricks) == (4, 4, 4)
    inv = scene.inverse(res)
    for i in range(2):
        ctx.set_calibration(i, scene.xyz[i], (16, 13, 16), scene.uv[i], (16, 13, 16), (0.5, 4.5))
        ctx.set_inverse_calibration(i, inv[i], res)
    ctx.step(scene.depth, scene.color)
    ref = orc.run_pipeline(scene, BMIN, BMAX, res, inv, limit=0.01, brick_size=g.brick_size,
                           bv=tuple(g.brick_voxels_axis), res_bricks=tuple(g.res_bricks))
    assert np.array_equal(ctx.readback_brick_counters(), ref["counters"])
    got = ctx.readback_tsdf()
    assert same_bits(got, ref["tsdf"]), "%d voxels differ" % count_diff(got, ref["tsdf"])
    assert 0 < len(ref["occupied"]) < g.num_bricks
    ctx.close()


def test_edge_cases(pkg, orc):
    scene, ctx, inv = build(pkg, wh=(64, 53), G=32, lut_res=(16, 13, 16))
    # (a) empty frame: every depth 0
    z = np.zeros_like(scene.depth)
    ctx.step(z, scene.color)
    ref = oracle_run(orc, scene, ctx, inv, depth_override=z)
    check_images(ctx, ref, 2)
    assert same_bits(ctx.readback_tsdf(), ref["tsdf"])
    assert len(ctx.get_occupied()[0]) == 0
    # (b) all-invalid inverse LUT -> whole volume -limit, bricks off
    bad = [np.full_like(a, -1.0) for a in inv]
    for i in range(2):
        ctx.set_inverse_calibration(i, bad[i], (32, 32, 32))
    ctx.set_use_bricks(False)
    ctx.step(scene.depth, scene.color)
    ref = oracle_run(orc, scene, ctx, bad, use_bricks=False)
    got = ctx.readback_tsdf()
    assert same_bits(got, ref["tsdf"])
    # (c) depth with NaN / inf / negative / huge values
    d = scene.depth.copy()
    d[0, 5, 5], d[0, 6, 6], d[1, 7, 7], d[1, 8, 8] = np.nan, np.inf, -3.0, 1e30
    for i in range(2):
        ctx.set_inverse_calibration(i, inv[i], (32, 32, 32))
    ctx.step(d, scene.color)
    ref = oracle_run(orc, scene, ctx, inv, depth_override=d, use_bricks=False)
    check_images(ctx, ref, 2)
    assert same_bits(ctx.readback_tsdf(), ref["tsdf"])
    # (d) the same with the raw depths fed to the bilateral pass (no hole filling in front of it):
    # non-finite centres and taps reach pre_depth / pre_quality themselves
    rng = np.random.default_rng(3)
    for val in (np.nan, np.inf, -np.inf, -3.0, 1e30, 0.0):
        for _ in range(12):
            d[rng.integers(0, 2), rng.integers(0, 53), rng.integers(0, 64)] = val
    d[0, 20:23, 30:33] = np.nan
    d[1, 30:32, 10:14] = np.inf
    ctx.use_processed_depths(False)
    ctx.step(d, scene.color)
    ref = oracle_run(orc, scene, ctx, inv, depth_override=d, use_bricks=False)
    check_images(ctx, ref, 2)
    assert same_bits(ctx.readback_tsdf(), ref["tsdf"])
    ctx.close()


@pytest.mark.parametrize("where", ["inverse", "forward"])
def test_calibration_volumes_with_non_finite_entries(pkg, orc, where):
    """NaN, infinities and huge values sprinkled into the inverse LUT (texture coordinates and depths the sweep addresses the
    frame windows with) or into cv_xyz / cv_uv (world positions, colour coordinates, camera position, frustum planes): a
    damaged calibration must not crash or hang anything, and both sweeps still equal the oracle, which follows GL's rules
    for such coordinates (NaN -> texel 0, clamp to edge) -- except right next to a damaged texel of a 1:1 inverse LUT, see below"""
    scene, ctx, inv = build(pkg, wh=(64, 53), G=32, lut_res=(16, 13, 16))
    rng = np.random.default_rng(17)
    specials = np.float32([np.nan, np.inf, -np.inf, 3e38, -3e38, 1e20, -7.5, 0.0, 1.0])

    def poison(a, count):
        a = np.array(a, dtype=np.float32, copy=True)
        flat = a.reshape(-1)
        flat[rng.integers(0, flat.size, count)] = rng.choice(specials, count)
        return a

    if where == "inverse":
        inv = [poison(a, 400) for a in inv]
        inv[1][10:12, 8:16, 4:20] = np.nan                 # a whole region, so that entire tiles see nothing but NaN
        inv[0][20:22, 0:8, 0:8, 2] = np.inf
        for i in range(2):
            ctx.set_inverse_calibration(i, inv[i], (32, 32, 32))
    else:
        class Damaged:
            pass
        d = Damaged()
        d.__dict__.update(scene.__dict__)
        d.xyz = [poison(a, 60) for a in scene.xyz]
        d.uv = [poison(a, 60) for a in scene.uv]
        scene = d
        for i in range(2):
            ctx.set_calibration(i, scene.xyz[i], (16, 13, 16), scene.uv[i], (16, 13, 16), (0.5, 4.5))
            assert np.array_equal(ctx.camera_position(i), orc.camera_pos(scene.xyz[i]), equal_nan=True)
    for bricks in (True, False):
        ctx.set_use_bricks(bricks)
        ctx.set_skip_background(not bricks)                 # the classifier reads the same planes
        ctx.step(scene.depth, scene.color)
        ref = oracle_run(orc, scene, ctx, inv, use_bricks=bricks)
        check_images(ctx, ref, 2)
        assert np.array_equal(ctx.readback_brick_counters(), ref["counters"])
        got = ctx.readback_tsdf()
        if where == "forward":
            assert same_bits(got, ref["tsdf"]), (where, bricks, count_diff(got, ref["tsdf"]))
        else:
            # A voxel centre of a 1:1 LUT hits its texel with interpolation weights of exactly 0, and the library reads
            # that texel alone (the grid layout).  GL -- and the oracle -- still evaluate a + 0 * (b - a) with the next
            # texel b along every axis: for finite b that IS a; for a NaN or an infinite b it is NaN.  So the two agree
            # wherever the 2 x 2 x 2 texels a lookup touches are finite (and everywhere on an undamaged LUT); next to a
            # damaged texel the library's voxel keeps its own, clean entry.
            fin = np.ones((32, 32, 32), bool)
            for a in inv:
                f = np.isfinite(a).all(axis=-1)
                n = f.copy()
                for ax in range(3):
                    n &= np.concatenate([np.take(n, range(1, 32), axis=ax), np.take(n, [31], axis=ax)], axis=ax)
                fin &= n
            assert 0.5 < fin.mean() < 0.99
            same = (got == ref["tsdf"]) | (np.isnan(got) & np.isnan(ref["tsdf"]))
            assert same[fin].all(), (where, bricks, int((~same[fin]).sum()))
            assert (~same).sum() > 0            # (the convention above is visible on this LUT)
    ctx.close()


def test_normal_and_quality_in_one_launch_or_two(pkg, orc):
    """process_textures runs pre_normal + pre_quality as one kernel; a host that asks for the per-pass timers
    ("normal" / "quality", NetKinectArray.cpp:381-414) gets the two separate ones.  Same images, same brick
    counters, bit for bit, also with non-finite depth_b values reaching both passes."""
    scene, ctx, inv = build(pkg, n=3, wh=(200, 150), G=64, lut_res=(32, 27, 32))
    d = scene.depth.copy()
    rng = np.random.default_rng(11)
    for val in (np.nan, np.inf, -np.inf, -3.0, 1e30, 0.0):
        for _ in range(20):
            d[rng.integers(0, 3), rng.integers(0, 150), rng.integers(0, 200)] = val
    d[1, 60:64, 90:95] = np.nan
    for processed in (True, False):
        ctx.use_processed_depths(processed)
        ref = oracle_run(orc, scene, ctx, inv, depth_override=d, processed=processed)
        got = {}
        for timers in (False, True):
            ctx.enable_timers(timers)
            ctx.set_timer_detail(2)
            ctx.step(d, scene.color)
            check_images(ctx, ref, 3)
            assert np.array_equal(ctx.readback_brick_counters(), ref["counters"])
            assert same_bits(ctx.readback_tsdf(), ref["tsdf"])
            got[timers] = [ctx.readback_image(w, i) for w in (6, 7) for i in range(3)]
        assert all(same_bits(a, b) for a, b in zip(got[False], got[True]))
    ctx.close()


def test_bricked_equals_full_sweep_on_occupied_bricks(pkg):
    scene, ctx, inv = build(pkg)
    ctx.step(scene.depth, scene.color)
    bricked = ctx.readback_tsdf()
    ctx.set_use_bricks(False)
    ctx.integrate()
    full = ctx.readback_tsdf()
    ids, _ = ctx.get_occupied()
    g = ctx.geo
    mask = np.zeros(g.num_bricks, bool)
    mask[ids] = True
    m3 = mask.reshape(g.res_bricks[2], g.res_bricks[1], g.res_bricks[0])
    vox = np.kron(m3, np.ones((g.brick_voxels,) * 3, bool))[: g.res_volume[2], : g.res_volume[1], : g.res_volume[0]]
    assert same_bits(bricked[vox], full[vox])
    assert np.all(bricked[~vox] == np.float32(-0.01))
    assert 0 < vox.mean() < 1
    ctx.close()


@pytest.mark.parametrize("count,G,inv_res", [(2, 64, None), (3, 64, None), (4, 64, (45, 50, 45)), (3, 50, (50, 50, 50))])
def test_slab_contexts_reproduce_the_whole_volume(pkg, count, G, inv_res):
    """1 GPU vs k slabs: voxels are independent, so the concatenated slabs are
    bit-identical to the whole volume"""
    scene, whole, inv = build(pkg, G=G, inv_res=inv_res)
    whole.step(scene.depth, scene.color)
    full = whole.readback_tsdf()
    parts = []
    for r in range(count):
        _, ctx, _ = build(pkg, G=G, inv_res=inv_res, slab_rank=r, slab_count=count)
        ctx.step(scene.depth, scene.color)
        g = ctx.geo
        part = ctx.readback_tsdf()
        assert part.shape[0] == g.slab_voxel_z1 - g.slab_voxel_z0
        v = ctx.device_tsdf()
        assert v.halo_layers == 1 and v.owned_layers == g.slab_tile_z1 - g.slab_tile_z0
        assert v.owned == v.base + v.layer_bytes
        parts.append(part)
        ctx.close()
    assert same_bits(np.concatenate(parts, axis=0), full)
    whole.close()


def test_halo_layers_alias_device_memory(pkg):
    """the halo views handed to torch.distributed alias the tile-linear slab"""
    import torch

    from rgbd_recon_amd import dist as rdist

    scene, ctx, inv = build(pkg, slab_rank=1, slab_count=2)
    ctx.step(scene.depth, scene.color)
    ctx.sync()
    v = ctx.device_tsdf()
    send_lo, send_hi, recv_lo, recv_hi = rdist.halo_views(v, torch.device("cuda:0"))
    g = ctx.geo
    part = ctx.readback_tsdf()
    # first owned tile layer, de-tiled, equals the first 8 voxel rows of the slab
    t = send_lo.cpu().numpy().reshape(g.tiles[1], g.tiles[0], 8, 8, 8)      # ty, tx, z, y, x
    lin = t.transpose(2, 0, 3, 1, 4).reshape(8, g.tiles[1] * 8, g.tiles[0] * 8)
    assert same_bits(lin[:, : g.res_volume[1], : g.res_volume[0]], part[:8])
    recv_lo.fill_(7.0)
    torch.cuda.synchronize()
    assert float(wrap_first(rdist, v)) == 7.0
    ctx.close()


def wrap_first(rdist, v):
    import torch

    return rdist.wrap_device_floats(v.base, 1, torch.device("cuda:0"))[0].item()


def test_inverse_lut_tiling_round_trip(pkg):
    scene, ctx, inv = build(pkg, G=32, wh=(64, 53), lut_res=(16, 13, 16))
    got = ctx.readback_inverse_calibration(1, 0, 32)
    assert np.array_equal(got[..., :3], inv[1][..., :3])
    ctx.close()


def test_synthetic_device_lut_matches_host_generator(pkg):
    """the device-side analytic LUT (benchmark support) agrees with the numpy one"""
    scene, ctx, inv = build(pkg, G=64)
    for i in range(2):
        ctx.synth_inverse_calibration(i, scene.pinhole(i))
        got = ctx.readback_inverse_calibration(i, 0, 64)[..., :3]
        ref = inv[i][..., :3]
        valid = (got[..., 0] >= 0) & (ref[..., 0] >= 0)
        assert np.mean(valid) > 0.2
        assert np.mean((got[..., 0] >= 0) != (ref[..., 0] >= 0)) < 2e-3      # frustum edge rounding only
        np.testing.assert_allclose(got[valid], ref[valid], atol=2e-5)
    ctx.close()


# ---------------------------------------------------------------------------
def test_call_order_errors(pkg):
    capi, synth = pkg.capi, pkg.synth
    scene = synth.Scene(1, 64, 53, lut_res=(16, 13, 16))
    ctx = capi.Context(capi.make_config(1, (64, 53), voxel_size=2.0 / 32, brick_size=0.5), 0)
    with pytest.raises(capi.RgbdrError) as e:
        ctx.process_textures()
    assert e.value.status == capi.ERR_STATE
    ctx.update(scene.depth, scene.color)
    with pytest.raises(capi.RgbdrError) as e:
        ctx.process_textures()                      # calibration missing
    assert e.value.status == capi.ERR_STATE
    ctx.set_calibration(0, scene.xyz[0], (16, 13, 16), scene.uv[0], (16, 13, 16), (0.5, 4.5))
    ctx.clear_occupied_bricks()
    ctx.process_textures()
    with pytest.raises(capi.RgbdrError) as e:
        ctx.integrate()                             # inverse calibration missing
    assert e.value.status == capi.ERR_STATE
    with pytest.raises(capi.RgbdrError) as e:
        ctx.set_inverse_calibration(3, scene.inverse((32, 32, 32))[0], (32, 32, 32))
    assert e.value.status == capi.ERR_OUT_OF_RANGE
    for res in ((0, 32, 32), (32, 32768 + 1, 32), (0xffffffff, 0xffffffff, 0xffffffff)):
        # a resolution nothing can be: refused before any size is computed from it (the data pointer is never read)
        with pytest.raises(capi.RgbdrError) as e:
            ctx.set_inverse_calibration(0, np.zeros(4, np.float32), res)
        assert e.value.status == capi.ERR_INVALID_ARGUMENT
        with pytest.raises(capi.RgbdrError) as e:
            ctx.set_calibration(0, np.zeros(3, np.float32), res, scene.uv[0], (16, 13, 16), (0.5, 4.5))
        assert e.value.status == capi.ERR_INVALID_ARGUMENT
        with pytest.raises(capi.RgbdrError) as e:
            ctx.set_calibration(0, scene.xyz[0], (16, 13, 16), np.zeros(2, np.float32), res, (0.5, 4.5))
        assert e.value.status == capi.ERR_INVALID_ARGUMENT
    ctx.set_inverse_calibration(0, scene.inverse((32, 32, 32))[0], (32, 32, 32))
    with pytest.raises(capi.RgbdrError) as e:
        ctx.integrate()                             # bricks on, but not updated yet
    assert e.value.status == capi.ERR_STATE
    ctx.update_occupied_bricks()
    ctx.integrate()
    with pytest.raises(capi.RgbdrError) as e:
        ctx.readback_image(99, 0)
    assert e.value.status == capi.ERR_OUT_OF_RANGE
    with pytest.raises(capi.RgbdrError) as e:
        ctx.load_calibration_files(0, inv="/nonexistent/file.cv_xyz_inv")
    assert e.value.status == capi.ERR_IO
    with pytest.raises(capi.RgbdrError):
        capi.Context(capi.make_config(1, (64, 53)), 99)     # device id out of range
    ctx.close()


def test_setters_resize_and_limits(pkg, orc):
    scene, ctx, inv = build(pkg, wh=(64, 53), G=32, lut_res=(16, 13, 16))
    ctx.step(scene.depth, scene.color)
    # setTsdfLimit / setMinVoxelsPerBrick / setBrickSize
    ctx.set_tsdf_limit(0.03)
    ctx.set_min_voxels_per_brick(3)
    ctx.set_brick_size(2.0 / 32 * 4)
    g = ctx.geo
    assert g.brick_voxels == 4 and tuple(g.res_bricks) == (8, 8, 8)
    ctx.step(scene.depth, scene.color)
    ref = oracle_run(orc, scene, ctx, inv)
    assert np.array_equal(ctx.readback_brick_counters(), ref["counters"])
    assert same_bits(ctx.readback_tsdf(), ref["tsdf"])
    # setVoxelSize reallocates: the 1:1 LUT no longer fits and must be set again
    ctx.set_voxel_size(2.0 / 48)
    assert tuple(ctx.geo.res_volume) == (48, 48, 48)
    ctx.clear_occupied_bricks()
    ctx.process_textures()
    ctx.update_occupied_bricks()
    with pytest.raises(pkg.capi.RgbdrError) as e:
        ctx.integrate()
    assert e.value.status == pkg.capi.ERR_STATE
    for i in range(2):
        ctx.set_inverse_calibration(i, inv[i], (32, 32, 32))     # now a generic-resolution LUT
    ctx.step(scene.depth, scene.color)
    ref = oracle_run(orc, scene, ctx, inv)
    assert same_bits(ctx.readback_tsdf(), ref["tsdf"])
    ctx.close()


def test_calibration_files_round_trip(pkg, orc, tmp_path):
    """LUT files in the reference's on-disk format feed the context directly"""
    scene, ctx, inv = build(pkg, wh=(64, 53), G=32, lut_res=(16, 13, 16))
    ctx.step(scene.depth, scene.color)
    a = ctx.readback_tsdf()
    capi = pkg.capi
    ctx2 = capi.Context(capi.make_config(2, (64, 53), voxel_size=2.0 / 32, brick_size=0.5), 0)
    for i in range(2):
        px, pu, pi = (str(tmp_path / ("s%d.%s" % (i, e))) for e in ("cv_xyz", "cv_uv", "cv_xyz_inv"))
        assert orc.lut_write(px, scene.xyz[i], 3) == 0
        assert orc.lut_write(pu, scene.uv[i], 2) == 0
        assert orc.lut_write(pi, inv[i], 4) == 0
        ctx2.load_calibration_files(i, px, pu, pi)
        assert np.array_equal(ctx2.camera_position(i), orc.camera_pos(scene.xyz[i]))
    ctx2.step(scene.depth, scene.color)
    assert same_bits(ctx2.readback_tsdf(), a)
    ctx.close()
    ctx2.close()


def test_calibration_volumes_are_visible_on_the_device(pkg):
    """rgbdr_device_calibration: what CalibVolumes hands the other drawing modes as texture units (getXYZVolumeUnits /
    getUVVolumeUnits), getVolumeRes and getDepthLimits -- the records that were uploaded, in place, per sensor"""
    import torch
    from rgbd_recon_amd import dist as rdist
    capi, synth = pkg.capi, pkg.synth
    dev = torch.device("cuda", 0)
    scene = synth.Scene(2, 64, 53, lut_res=(16, 13, 16))
    ctx = capi.Context(capi.make_config(2, (64, 53), voxel_size=2.0 / 32, brick_size=0.5), 0)
    with pytest.raises(capi.RgbdrError) as e:
        ctx.device_calibration(0)
    assert e.value.status == capi.ERR_STATE
    ctx.set_calibration(0, scene.xyz[0], (16, 13, 16), scene.uv[0], (16, 13, 16), (0.5, 4.5))
    ctx.set_calibration(1, scene.xyz[1], (16, 13, 16), scene.uv[1], (16, 13, 16), (0.25, 5.5))
    with pytest.raises(capi.RgbdrError) as e:
        ctx.device_calibration(2)
    assert e.value.status == capi.ERR_OUT_OF_RANGE
    inv = scene.inverse((32, 32, 32))
    ctx.set_inverse_calibration(1, inv[1], (32, 32, 32))
    for i in range(2):
        v = ctx.device_calibration(i)
        assert list(v.xyz_res) == [16, 13, 16] and list(v.uv_res) == [16, 13, 16]
        assert list(v.inv_res) == ([32, 32, 32] if i == 1 else [0, 0, 0])
        assert tuple(v.depth_limits) == ((0.5, 4.5), (0.25, 5.5))[i]
        n = 16 * 13 * 16
        xyz = rdist.wrap_device_floats(v.cv_xyz, n * 4, dev).cpu().numpy().reshape(16, 13, 16, 4)
        uv = rdist.wrap_device_floats(v.cv_uv, n * 2, dev).cpu().numpy().reshape(16, 13, 16, 2)
        assert same_bits(xyz[..., :3], scene.xyz[i]) and same_bits(uv, scene.uv[i])
        assert v.stream == ctx.stream()
    ctx.close()


def test_contexts_driven_by_concurrent_threads(pkg, orc):
    """a context is single-caller, but a process may drive several (one per slab, one per client): four threads create their
    own contexts at the same time and run frames of different scenes, grids, limits, sweeps and schedules next to each other on
    one GPU (ctypes releases the GIL around every call); each volume, image set and brick table equals the oracle's --
    nothing is shared between contexts except read-only tables"""
    import threading
    jobs = [dict(n=2, G=64, seed=1234, tsdf_limit=0.03, bricks=True, pipelined=False),
            dict(n=3, G=48, seed=7, tsdf_limit=0.05, bricks=False, pipelined=True),
            dict(n=1, G=32, seed=99, tsdf_limit=0.02, bricks=True, pipelined=True),
            dict(n=4, G=40, seed=5, tsdf_limit=0.04, bricks=False, pipelined=False)]
    results, errors = [None] * len(jobs), []
    start = threading.Barrier(len(jobs))

    def worker(k, job):
        try:
            start.wait()
            scene, ctx, inv = build(pkg, n=job["n"], G=job["G"], seed=job["seed"], tsdf_limit=job["tsdf_limit"])
            other = pkg.synth.Scene(job["n"], 128, 106, lut_res=(32, 27, 32), seed=job["seed"] + 1, sphere_r=0.7)
            ctx.set_use_bricks(job["bricks"])
            ctx.set_pipelined(job["pipelined"])
            for rep in range(6):                                  # frames alternate; the last one is `scene`
                sc = other if rep % 2 == 0 else scene
                md, mc = ctx.map_frame_buffer()
                md[:] = sc.depth.view(np.uint8).reshape(-1)
                mc[:] = sc.color.reshape(-1)
                ctx.upload_mapped_frame()
                ctx.clear_occupied_bricks()
                ctx.process_textures()
                ctx.update_occupied_bricks()
                ctx.integrate()
            results[k] = (scene, ctx, inv, ctx.readback_tsdf(), ctx.readback_brick_counters(),
                          [[ctx.readback_image(w, i) for i in range(job["n"])] for w in IMG.values()])
        except BaseException as e:                                # noqa: reported by the main thread
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=worker, args=(k, j)) for k, j in enumerate(jobs)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
    assert not errors, errors
    for k, job in enumerate(jobs):
        scene, ctx, inv, tsdf, counters, images = results[k]
        ref = oracle_run(orc, scene, ctx, inv, use_bricks=job["bricks"])
        assert same_bits(tsdf, ref["tsdf"]), (k, count_diff(tsdf, ref["tsdf"]))
        assert np.array_equal(counters, ref["counters"]), k
        for name, per_sensor in zip(IMG, images):
            for i, got in enumerate(per_sensor):
                assert same_bits(got, ref[name][i]), (k, name, i)
        ctx.close()


def test_create_destroy_cycles_return_every_byte(pkg):
    """contexts come and go (a host that re-creates its backend when the scene changes): 25 cycles that touch every optional
    allocation -- LUT arena, view / peel / fill buffers, page-locked double buffer, halo staging sets and stream, timers, skip
    tables, DXT staging -- destroyed with work still queued, leave the device's free memory where it was"""
    import torch
    capi, synth = pkg.capi, pkg.synth
    scene = synth.Scene(2, 64, 53, lut_res=(16, 13, 16))
    inv = scene.inverse((48, 48, 48))
    view = capi.make_view((2.2, 1.6, 1.9), (0.0, 0.9, 0.0), (0.0, 1.0, 0.0), 50.0, 64, 48, BMIN, BMAX)

    def cycle(k):
        slab = dict(slab_rank=k % 2, slab_count=2) if k % 3 == 0 else {}
        ctx = capi.Context(capi.make_config(2, (64, 53), voxel_size=2.0 / 48, brick_size=8 * 2.0 / 48, **slab), 0)
        for i in range(2):
            ctx.set_calibration(i, scene.xyz[i], (16, 13, 16), scene.uv[i], (16, 13, 16), (0.5, 4.5))
            ctx.set_inverse_calibration(i, inv[i], (48, 48, 48))
        ctx.enable_timers(True)
        ctx.set_skip_background(k % 2 == 0)
        ctx.set_pipelined(k % 4 == 1)
        if slab:
            for b in range(2):
                ctx.halo_staging(b)
            ctx.halo_begin_step()
        md, mc = ctx.map_frame_buffer()
        md[:] = scene.depth.view(np.uint8).reshape(-1)
        mc[:] = scene.color.reshape(-1)
        ctx.upload_mapped_frame()
        ctx.clear_occupied_bricks()
        ctx.process_textures()
        ctx.update_occupied_bricks()
        ctx.integrate()
        if not slab:
            view.skip_space = k % 2
            ctx.raymarch(view)
            ctx.fill_colors(64, 48)
            ctx.draw_depth_limits(view)
        ctx.step(scene.depth, scene.color)                   # left queued: destroy has to drain it
        ctx.close()

    cycle(0)
    cycle(1)                                                 # (first uses may keep runtime-internal pools)
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for k in range(25):
        cycle(k)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 8 << 20, "device memory shrank by %.1f MiB over 25 create / destroy cycles" % ((free0 - free1) / 2 ** 20)


def test_damaged_calibration_files_are_io_errors(pkg, orc, tmp_path):
    """a LUT file is sized by its own header: a truncated payload, a header that promises more than the file holds (up to
    2^96 records: nothing is allocated on its word), a zero resolution and a file shorter than the header all come back as
    RGBDR_ERR_IO -- the process lives, the context keeps the calibration it had and still produces the same volume"""
    import struct
    capi = pkg.capi
    scene, ctx, inv = build(pkg, wh=(64, 53), G=32, lut_res=(16, 13, 16))
    ctx.step(scene.depth, scene.color)
    want = ctx.readback_tsdf()
    good = {}
    for ext, data, comps in (("cv_xyz", scene.xyz[0], 3), ("cv_uv", scene.uv[0], 2), ("cv_xyz_inv", inv[0], 4)):
        good[ext] = str(tmp_path / ("good." + ext))
        assert orc.lut_write(good[ext], data, comps) == 0
    ctx.load_calibration_files(0, good["cv_xyz"], good["cv_uv"], good["cv_xyz_inv"])       # the files themselves are fine
    def variants(raw):
        res = struct.unpack("<3I", raw[:12])
        return {
            "payload cut in half": raw[:20 + (len(raw) - 20) // 2],
            "one byte short": raw[:-1],
            "header only": raw[:20],
            "shorter than the header": raw[:7],
            "empty": b"",
            "resolution 0": struct.pack("<3I", 0, res[1], res[2]) + raw[12:],
            "resolution 2^32 - 1 on every axis": struct.pack("<3I", 0xffffffff, 0xffffffff, 0xffffffff) + raw[12:],
            "one row too many": struct.pack("<3I", res[0], res[1] + 1, res[2]) + raw[12:],
        }

    path = str(tmp_path / "bad.lut")
    for ext in ("cv_xyz_inv", "cv_xyz", "cv_uv"):
        damaged = variants(open(good[ext], "rb").read())
        for what, blob in damaged.items():
            open(path, "wb").write(blob)
            with pytest.raises(capi.RgbdrError) as e:
                if ext == "cv_xyz_inv":
                    ctx.load_calibration_files(0, inv=path)
                elif ext == "cv_xyz":
                    ctx.load_calibration_files(0, path, good["cv_uv"], None)
                else:
                    ctx.load_calibration_files(0, good["cv_xyz"], path, None)
            assert e.value.status == capi.ERR_IO, (ext, what)
    open(path, "wb").write(damaged["resolution 2^32 - 1 on every axis"])
    with pytest.raises(capi.RgbdrError) as e:
        ctx.load_calibration_files(0, good["cv_xyz"], path, None)
    assert "header says 4294967295 x 4294967295 x 4294967295 records of 8 bytes" in str(e.value)
    ctx.step(scene.depth, scene.color)
    assert same_bits(ctx.readback_tsdf(), want)
    ctx.close()


def test_timers_report_each_pass(pkg, orc):
    scene, ctx, inv = build(pkg)
    ctx.enable_timers(True)
    ctx.step(scene.depth, scene.color)
    ctx.sync()
    names = ["morph", "bilateral", "boundary", "normal", "quality", "1preprocess", "2integrate", "bricks"]
    t = {n: ctx.timer_ns(n) for n in names}
    assert all(v > 0 for v in t.values())
    assert t["1preprocess"] >= t["bilateral"]
    # the timers never change a result, whatever their level of detail
    ref = oracle_run(orc, scene, ctx, inv)
    for detail in (2, 1):
        ctx.set_timer_detail(detail)
        ctx.step(scene.depth, scene.color)
        check_images(ctx, ref, 2)
        assert np.array_equal(ctx.readback_brick_counters(), ref["counters"])
        assert same_bits(ctx.readback_tsdf(), ref["tsdf"])
    ctx.set_timer_detail(2)
    with pytest.raises(pkg.capi.RgbdrError):
        ctx.timer_ns("nope")
    ctx.close()


def test_accumulating_timers_count_every_interval_however_long_nobody_asks(pkg):
    """rgbdr_enable_timer_accumulation keeps an event pair per interval until rgbdr_timer_stats reads them; a host that
    lets it run keeps at most a few thousand pairs (older intervals are folded into a sum): 5000 frames without a
    question, then every interval is accounted for exactly once"""
    scene, ctx, _ = build(pkg, wh=(64, 53), G=32, lut_res=(16, 13, 16))
    ctx.enable_timer_accumulation(True)
    ctx.set_timer_detail(1)
    frames = 5000
    for _ in range(frames):
        ctx.update(scene.depth, scene.color) if _ == 0 else None
        ctx.clear_occupied_bricks()
        ctx.process_textures()
        ctx.update_occupied_bricks()
        ctx.integrate()
    total, count = ctx.timer_stats("2integrate")
    assert count == frames and total > frames * 1000, (count, total)          # (a sweep of 32^3 takes a few microseconds)
    total_p, count_p = ctx.timer_stats("1preprocess")
    assert count_p == frames and total_p > total_p // frames > 0
    assert ctx.timer_stats("2integrate") == (0, 0)                            # read once
    ctx.integrate()
    assert ctx.timer_stats("2integrate")[1] == 1
    ctx.close()


def test_device_resident_frames(pkg):
    import torch

    scene, ctx, inv = build(pkg)
    ctx.step(scene.depth, scene.color)
    a = ctx.readback_tsdf()
    d = torch.from_numpy(scene.depth).cuda()
    c = torch.from_numpy(scene.color).cuda()
    torch.cuda.synchronize()
    ctx.update_device(d.data_ptr(), c.data_ptr())
    ctx.clear_occupied_bricks()
    ctx.process_textures()
    ctx.update_occupied_bricks()
    ctx.integrate()
    assert same_bits(ctx.readback_tsdf(), a)
    ctx.close()


def test_external_stream(pkg):
    """rgbdr_set_stream: the context enqueues on a torch stream, results unchanged"""
    import torch

    scene, ctx, inv = build(pkg)
    ctx.step(scene.depth, scene.color)
    a = ctx.readback_tsdf()
    s = torch.cuda.Stream()
    ctx.set_stream(s.cuda_stream)
    assert ctx.stream() == s.cuda_stream
    ctx.step(np.zeros_like(scene.depth), scene.color)
    ctx.step(scene.depth, scene.color)
    s.synchronize()
    assert same_bits(ctx.readback_tsdf(), a)
    ctx.set_stream(None)
    assert ctx.stream() != s.cuda_stream
    ctx.step(scene.depth, scene.color)
    assert same_bits(ctx.readback_tsdf(), a)
    ctx.close()


def test_device_images_alias_what_readback_returns(pkg):
    """rgbdr_device_image: zero-copy views for consumers that stay on the device (the textures the other
    Reconstructions bind, recon_trigrid.cpp:30-33) -- every view is the memory rgbdr_readback_image copies from"""
    import torch

    from rgbd_recon_amd import dist as rdist

    capi = pkg.capi
    scene, ctx, inv = build(pkg)
    ctx.step(scene.depth, scene.color)
    ctx.sync()
    dev = torch.device("cuda", 0)
    for which in range(8):
        for sensor in range(2):
            v = ctx.device_image(which, sensor)
            ch = capi.IMG_CHANNELS[which]
            assert (v.width, v.height, v.channels, v.element_bytes) == (128, 106, ch, 4) and v.stream
            t = rdist.wrap_device_floats(v.ptr, 128 * 106 * ch, dev).cpu().numpy().reshape(106, 128, ch)
            want = ctx.readback_image(which, sensor).reshape(106, 128, ch)
            assert same_bits(t, want), (which, sensor)
    v = ctx.device_image(capi.IMG_COLOR, 1)
    assert (v.width, v.height, v.channels, v.element_bytes) == (128, 106, 3, 1)
    class U8:
        __cuda_array_interface__ = {"shape": (128 * 106 * 3,), "typestr": "|u1", "data": (int(v.ptr), False), "version": 2}

    raw = torch.as_tensor(U8(), device=dev)
    assert np.array_equal(raw.cpu().numpy().reshape(106, 128, 3), ctx.readback_color(1))
    # writing through the view is visible to the next consumer of the image (it IS the texture)
    q = ctx.device_image(capi.IMG_QUALITY, 0)
    rdist.wrap_device_floats(q.ptr, 128 * 106, dev).fill_(0.25)
    torch.cuda.synchronize()
    assert np.all(ctx.readback_image(capi.IMG_QUALITY, 0) == np.float32(0.25))
    with pytest.raises(capi.RgbdrError):
        ctx.device_image(9, 0)
    with pytest.raises(capi.RgbdrError):
        ctx.device_image(0, 2)
    ctx.close()


def test_inverse_luts_of_mixed_resolutions_in_one_context(pkg, orc):
    """sensor 0 with a LUT 1:1 with the grid, sensor 1 with a coarser one (both end up in the grid layout);
    and with RGBDR_FLAG_NO_RESAMPLE both stay in the file layout -- each against the oracle, which samples the
    LUTs at their own resolutions"""
    capi, synth = pkg.capi, pkg.synth
    G = 64
    for flags in (capi.FLAGS_DEFAULT, capi.FLAGS_DEFAULT | capi.FLAG_NO_RESAMPLE):
        scene = synth.Scene(2, 128, 106, lut_res=(32, 27, 32), seed=1234)
        ctx = capi.Context(capi.make_config(2, (128, 106), voxel_size=2.0 / G, brick_size=8 * 2.0 / G, flags=flags), 0)
        inv = [synth.inverse_lut(scene.sensors[0], (G, G, G)), synth.inverse_lut(scene.sensors[1], (45, 50, 41))]
        for i in range(2):
            ctx.set_calibration(i, scene.xyz[i], (32, 27, 32), scene.uv[i], (32, 27, 32), (0.5, 4.5))
        ctx.set_inverse_calibration(0, inv[0], (G, G, G))
        ctx.set_inverse_calibration(1, inv[1], (45, 50, 41))
        for bricks in (True, False):
            ctx.set_use_bricks(bricks)
            ctx.step(scene.depth, scene.color)
            ref = oracle_run(orc, scene, ctx, inv, use_bricks=bricks)
            assert same_bits(ctx.readback_tsdf(), ref["tsdf"]), (flags, bricks, count_diff(ctx.readback_tsdf(), ref["tsdf"]))
        # replacing one sensor's LUT by another resolution afterwards keeps working
        inv[0] = synth.inverse_lut(scene.sensors[0], (50, 64, 37))
        ctx.set_inverse_calibration(0, inv[0], (50, 64, 37))
        ctx.step(scene.depth, scene.color)
        ref = oracle_run(orc, scene, ctx, inv, use_bricks=False)
        assert same_bits(ctx.readback_tsdf(), ref["tsdf"])
        ctx.close()


def test_pipelined_mode_is_schedule_only(pkg):
    """RGBDR_FLAG_PIPELINE overlaps the pre_* chain of frame k+1 with integrate of
    frame k on two streams with double-buffered frames/masks; results per frame are
    identical to the sequential schedule"""
    scene, ctx, inv = build(pkg)
    rng = np.random.default_rng(3)
    frames = [scene.depth, np.zeros_like(scene.depth), np.roll(scene.depth, 7, axis=2),
              (scene.depth * (rng.random(scene.depth.shape) > 0.1)).astype(np.float32), scene.depth]
    want, occ = [], []
    for f in frames:
        ctx.step(f, scene.color)
        want.append(ctx.readback_tsdf())
        occ.append(ctx.get_occupied()[0])
    want_quality = ctx.readback_image(7, 1)          # quality image of the last frame, sequential schedule
    want_depth_b = ctx.readback_image(4, 0)
    ctx.set_pipelined(True)
    for rounds in range(2):
        for k, f in enumerate(frames):
            ctx.step(f, scene.color)
            if rounds == 0:                      # readback after every frame
                assert same_bits(ctx.readback_tsdf(), want[k]), k
                assert np.array_equal(ctx.get_occupied()[0], occ[k])
        # second round: frames queued back to back, only the last one is inspected
    assert same_bits(ctx.readback_tsdf(), want[-1])
    assert same_bits(ctx.readback_image(7, 1), want_quality) and same_bits(ctx.readback_image(4, 0), want_depth_b)
    # the brick-skipping sweep reads the right mask buffer
    ctx.set_pipelined(False)
    ctx.step(frames[2], scene.color)
    assert same_bits(ctx.readback_tsdf(), want[2])
    ctx.close()


@pytest.mark.parametrize("mode", [1, 5])
def test_dxt_compressed_colour_frames(pkg, orc, mode):
    """compress_rgb 1 (DXT1, the reference's yml default) and 5 (DXT5): frames are
    decoded on the device exactly as the reference's CPU decoder (squish) does"""
    capi, synth = pkg.capi, pkg.synth
    scene, ctx, inv = build(pkg, compress_rgb=mode)
    W, H = 128, 106
    blocks = np.stack([synth.encode_dxt(scene.color[i], mode) for i in range(2)])
    rng = np.random.default_rng(9)
    blocks[1, : blocks.shape[1] // 4] = rng.integers(0, 256, blocks.shape[1] // 4, dtype=np.uint8)   # arbitrary blocks too
    ctx.step(scene.depth, blocks)
    decoded = np.stack([orc.decode_dxt(blocks[i], W, H, mode) for i in range(2)])
    for i in range(2):
        assert np.array_equal(ctx.readback_color(i), decoded[i])
    assert np.mean(np.abs(decoded[0].astype(int) - scene.color[0].astype(int))) < 6      # the encoder is sane

    class Decoded:
        pass

    s2 = Decoded()
    s2.__dict__.update(scene.__dict__)
    s2.color = decoded
    ref = oracle_run(orc, s2, ctx, inv)
    check_images(ctx, ref, 2)
    assert same_bits(ctx.readback_tsdf(), ref["tsdf"])
    # the pre_* chain read the colour above from the RGB8 frame (readback_color had it decoded); a fresh upload is
    # read straight from its DXT blocks, and a consumer of the RGB8 image gets it decoded then
    ctx.step(scene.depth, blocks)
    check_images(ctx, ref, 2)
    assert same_bits(ctx.readback_tsdf(), ref["tsdf"])
    view = ctx.device_image(8, 1)
    assert view.width == W and view.height == H and view.channels == 3
    assert np.array_equal(ctx.readback_color(1), decoded[1])
    ctx.process_textures()
    ctx.clear_occupied_bricks()
    ctx.step(scene.depth, blocks)
    check_images(ctx, ref, 2)
    ctx.close()


def test_lut_arena_placement_probe(pkg, monkeypatch):
    """placing the LUT arena by timing candidate allocations: RGBDR_ARENA_TRIALS=n for arenas of 256 MiB and more; unset,
    arenas below 1 GiB (this one) take the first allocation without probing, larger ones try up to sixteen (the library's default).  The choice
    never changes a result, and a probe leaves the volume cleared and marked as not integrated."""
    out = []
    for trials in (None, "1", "4"):
        if trials:
            monkeypatch.setenv("RGBDR_ARENA_TRIALS", trials)
        else:
            monkeypatch.delenv("RGBDR_ARENA_TRIALS", raising=False)
        scene, ctx, _ = build(pkg, n=2, G=256, lut_res=(32, 27, 32))         # 2 x 256^3 x 12 B = 403 MB arena
        ms, kept = ctx.arena_probe()
        if trials == "4":
            assert 1 <= len(ms) <= 4 and 0 <= kept < len(ms) and all(m > 0 for m in ms)
            assert ms[kept] == min(ms)
            view = pkg.capi.make_view((2.4, 1.8, 2.1), (0.0, 0.7, 0.0), (0.0, 1.0, 0.0), 45.0, 32, 24, BMIN, BMAX)
            with pytest.raises(pkg.capi.RgbdrError):                          # the probe's replay wrote into the volume
                ctx.raymarch(view)
        else:
            assert ms == [0.0] and kept == 0                                  # default and "1": no probing
        ctx.set_use_bricks(False)
        ctx.step(scene.depth, scene.color)
        out.append(ctx.readback_tsdf())
        ctx.close()
    assert same_bits(out[0], out[1]) and same_bits(out[0], out[2])
    monkeypatch.setenv("RGBDR_ARENA_TRIALS", "4")
    small = build(pkg)[1]                                                     # 64^3: below the threshold
    assert small.arena_probe() == ([0.0], 0)
    small.close()


@pytest.mark.parametrize("slab", [None, (1, 3)])
def test_lut_arena_assembled_from_the_fastest_chunks(pkg, orc, monkeypatch, slab):
    """The LUT arena as a range of mapped physical chunks (rgbdr_get_arena_chunks; forced here with 64-MiB chunks -- the
    library does it on its own when no candidate allocation streams at the fast level): results are the oracle's bit for
    bit, a resize re-assembles it, and destroy returns every byte."""
    import torch

    monkeypatch.setenv("RGBDR_ARENA_CHUNKS", "force")
    monkeypatch.setenv("RGBDR_ARENA_CHUNK_MB", "64")
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    kw = dict(slab_rank=slab[0], slab_count=slab[1]) if slab else {}
    scene, ctx, inv = build(pkg, n=3, G=192, lut_res=(32, 27, 32), **kw)         # 3 x 192^3 x 12 B = 255 MB (a third of it per slab)
    chunks, ms = ctx.arena_chunks()
    g = ctx.geo
    layers = (g.slab_tile_z1 - g.slab_tile_z0) + 2 * g.halo_tile_layers
    arena = int(g.tiles[0]) * int(g.tiles[1]) * layers * 3 * 3 * 512 * 4
    assert chunks == -(-arena // (64 << 20)) and ms > 0.0
    whole = build(pkg, n=3, G=192, lut_res=(32, 27, 32))[1] if slab else ctx
    for bricks in (False, True):
        ctx.set_use_bricks(bricks)
        ctx.step(scene.depth, scene.color)
        got = ctx.readback_tsdf()
        ref = oracle_run(orc, scene, whole, inv, use_bricks=bricks)["tsdf"]
        if slab:
            ref = ref[g.slab_voxel_z0:g.slab_voxel_z1]
        assert same_bits(got, ref), "bricks %d: %d voxels differ" % (bricks, count_diff(got, ref))
    if slab:
        whole.close()
    if not slab:                                                              # setVoxelSize: the arena is assembled anew
        ctx.set_voxel_size(2.0 / 160)
        ctx.set_brick_size(8 * 2.0 / 160)
        inv2 = scene.inverse((160, 160, 160))
        for i in range(3):
            ctx.set_inverse_calibration(i, inv2[i], (160, 160, 160))
        assert ctx.arena_chunks()[0] == -(-(160 // 8) ** 3 * 3 * 3 * 512 * 4 // (64 << 20))
        ctx.set_use_bricks(False)
        ctx.step(scene.depth, scene.color)
        assert same_bits(ctx.readback_tsdf(), oracle_run(orc, scene, ctx, inv2)["tsdf"])
    ctx.close()
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] >= free0 - (8 << 20), "the chunks of the arena were not returned"
    monkeypatch.setenv("RGBDR_ARENA_CHUNKS", "0")
    plain = build(pkg, n=3, G=192, lut_res=(32, 27, 32))[1]
    assert plain.arena_chunks() == (0, 0.0)
    plain.close()


def test_brick_sweep_skips_tiles_that_are_still_cleared(pkg, orc):
    """the brick-skipping sweep rewrites -limit only where a tile does not hold it already;
    every event that invalidates that record is followed by a correct volume"""
    scene, ctx, inv = build(pkg, G=64)
    scene2 = pkg.synth.Scene(2, 128, 106, lut_res=(32, 27, 32), seed=4321, sphere_r=0.6)

    def expect(sc, **kw):
        return oracle_run(orc, sc, ctx, inv, **kw)["tsdf"]

    ctx.step(scene.depth, scene.color)
    a = expect(scene)
    assert same_bits(ctx.readback_tsdf(), a)
    ctx.step(scene.depth, scene.color)                          # steady state: nothing to clear
    assert same_bits(ctx.readback_tsdf(), a)
    ctx.step(scene2.depth, scene2.color)                        # the surface moved: old tiles must be cleared
    b = expect(scene2)
    assert same_bits(ctx.readback_tsdf(), b) and not same_bits(a, b)
    ctx.set_use_bricks(False)                                   # a full sweep writes every tile ...
    ctx.integrate()
    assert same_bits(ctx.readback_tsdf(), expect(scene2, use_bricks=False))
    ctx.set_use_bricks(True)                                    # ... so the next brick sweep clears again
    ctx.integrate()
    assert same_bits(ctx.readback_tsdf(), b)
    ctx.set_tsdf_limit(0.02)                                    # another -limit
    ctx.integrate()
    assert same_bits(ctx.readback_tsdf(), expect(scene2, limit=np.float32(0.02)))
    ctx.set_tsdf_limit(0.01)
    ctx.step(scene.depth, scene.color)
    assert same_bits(ctx.readback_tsdf(), a)
    ctx.close()


def test_mapped_frame_buffers_equal_plain_upload(pkg):
    """the page-locked double frame buffer (the reference's double_pbo) feeds the same frames"""
    scene, ctx, _ = build(pkg)
    scene2 = pkg.synth.Scene(2, 128, 106, lut_res=(32, 27, 32), seed=77, sphere_r=0.7)
    want = []
    for sc in (scene, scene2, scene):
        ctx.step(sc.depth, sc.color)
        want.append(ctx.readback_tsdf())
    got = []
    for sc in (scene, scene2, scene):                      # three frames: both buffers get reused
        d, c = ctx.map_frame_buffer()
        assert d.nbytes == sc.depth.nbytes and c.nbytes == sc.color.nbytes
        d[:] = sc.depth.view(np.uint8).reshape(-1)
        c[:] = sc.color.reshape(-1)
        ctx.upload_mapped_frame()
        ctx.clear_occupied_bricks()
        ctx.process_textures()
        ctx.update_occupied_bricks()
        ctx.integrate()
        got.append(ctx.readback_tsdf())
    for a, b in zip(want, got):
        assert same_bits(a, b)
    assert not same_bits(want[0], want[1])
    fresh = build(pkg)[1]
    with pytest.raises(pkg.capi.RgbdrError) as e:
        fresh.upload_mapped_frame()
    assert e.value.status == pkg.capi.ERR_STATE
    fresh.close()
    ctx.close()


def test_host_frames_queued_back_to_back_without_synchronisation(pkg):
    """Host uploads ride on their own copy stream into two device staging sets (NetKinectArray::update's PBO pair,
    NetKinectArray.cpp:226-238) while earlier frames are still being processed: a run of different frames enqueued
    with no host synchronisation in between -- pageable uploads, the page-locked double buffer, both schedules, the
    brick sweep and the full sweep, u8 depth + DXT1 colour as well -- must leave exactly what the same frames give
    one at a time (every frame's volume is checked: the staging set of frame k is refilled by frame k + 2)."""
    capi, synth = pkg.capi, pkg.synth
    for cfgkw in ({}, {"compress_depth": 1, "compress_rgb": 1}):
        scene, ctx, _ = build(pkg, **cfgkw)
        frames = []
        for seed, r in ((1234, 0.9), (77, 0.7), (5, 0.8), (901, 0.6), (33, 0.75)):
            sc = synth.Scene(2, 128, 106, lut_res=(32, 27, 32), seed=seed, sphere_r=r)
            depth = synth.compress_depth_u8(sc.depth) if cfgkw else sc.depth
            color = np.stack([synth.encode_dxt(sc.color[i], 1) for i in range(2)]) if cfgkw else sc.color
            frames.append((np.ascontiguousarray(depth), np.ascontiguousarray(color)))
        want = []
        for d, c in frames:                                   # one at a time, drained after each
            ctx.step(d, c)
            ctx.sync()
            want.append(ctx.readback_tsdf())
        assert not same_bits(want[0], want[1])
        for pipelined in (False, True):
            for bricks in (True, False):
                ctx.set_pipelined(pipelined)
                ctx.set_use_bricks(bricks)
                ref = []
                for d, c in frames:
                    ctx.step(d, c)
                    ref.append(ctx.readback_tsdf())
                for mapped in (False, True):
                    views = []
                    for k, (d, c) in enumerate(frames * 3):       # 15 frames in flight behind each other
                        if mapped:
                            md, mc = ctx.map_frame_buffer()
                            md[:] = d.view(np.uint8).reshape(-1)
                            mc[:] = c.view(np.uint8).reshape(-1)
                            ctx.upload_mapped_frame()
                        else:
                            ctx.update(d, c)
                        ctx.clear_occupied_bricks()
                        ctx.process_textures()
                        ctx.update_occupied_bricks()
                        ctx.integrate()
                        if k >= 10:                                # the last five: read back (stream-ordered) and compare
                            views.append(ctx.readback_tsdf())
                    for k, v in enumerate(views):
                        assert same_bits(v, ref[k]), "pipelined %s bricks %s mapped %s frame %d: %d voxels differ" % (
                            pipelined, bricks, mapped, k, count_diff(v, ref[k]))
                if bricks and not pipelined:
                    for a, b in zip(ref, want):
                        assert same_bits(a, b)
        ctx.set_pipelined(False)
        ctx.close()


def test_upload_frame_has_consumed_a_page_locked_source_when_it_returns(pkg):
    """rgbdr_upload_frame copies its inputs (NetKinectArray::update memcpys the message into the PBO,
    NetKinectArray.cpp:533-535): the caller may overwrite its buffers as soon as the call returns -- also when it
    page-locked them itself, in which case the DMA out of them is asynchronous and the library waits for it"""
    import torch
    scene, ctx, _ = build(pkg)
    scene2 = pkg.synth.Scene(2, 128, 106, lut_res=(32, 27, 32), seed=77, sphere_r=0.7)
    ctx.step(scene.depth, scene.color)
    want = ctx.readback_tsdf()
    d = torch.empty(scene.depth.shape, dtype=torch.float32).pin_memory()
    c = torch.empty(scene.color.shape, dtype=torch.uint8).pin_memory()
    for _ in range(5):
        d.copy_(torch.from_numpy(scene.depth))
        c.copy_(torch.from_numpy(scene.color))
        ctx.update(d.numpy(), c.numpy())
        d.copy_(torch.from_numpy(scene2.depth))               # scribble over the source at once
        c.copy_(torch.from_numpy(scene2.color))
        ctx.clear_occupied_bricks()
        ctx.process_textures()
        ctx.update_occupied_bricks()
        ctx.integrate()
        assert same_bits(ctx.readback_tsdf(), want)
    ctx.close()


def test_settle_leaves_a_usable_context(pkg, orc):
    """rgbdr_settle scribbles over the volume by design; the next sweeps (brick-skipping
    included, whose clear-skipping must not trust the scribbled tiles) are correct again"""
    scene, ctx, inv = build(pkg)
    ctx.step(scene.depth, scene.color)
    want = ctx.readback_tsdf()
    assert ctx.settle(0.5) > 0.0
    ctx.integrate()
    assert same_bits(ctx.readback_tsdf(), want)
    ctx.set_use_bricks(False)
    ctx.integrate()
    full = ctx.readback_tsdf()
    ctx.settle(0.2)
    ctx.integrate()
    assert same_bits(ctx.readback_tsdf(), full)
    ctx.close()


@pytest.mark.parametrize("n,G,wh", [(2, 64, (128, 106)), (5, 64, (128, 106)), (3, 96, (200, 150)), (1, 32, (64, 53)),
                                    (4, 64, (128, 106)), (8, 64, (128, 106)), (7, 40, (96, 80))])
def test_full_sweep_background_skip(pkg, orc, n, G, wh):
    """RGBDR_FLAG_SKIP_BACKGROUND leaves the LUT planes of (tile, sensor) pairs unread whose window shows only
    background; the volume is the oracle's bit for bit, with and without the flag, also next to store elision,
    for frames with holes / non-finite depths, an empty frame and a frame that is all surface."""
    scene, ctx, inv = build(pkg, n=n, wh=wh, G=G)
    ctx.set_use_bricks(False)
    rng = np.random.default_rng(5)
    frames = [scene.depth.copy()]
    d = scene.depth.copy()
    for val in (np.nan, np.inf, -np.inf, -3.0, 1e30, 0.0):
        for _ in range(16):
            d[rng.integers(0, n), rng.integers(0, wh[1]), rng.integers(0, wh[0])] = val
    frames.append(d)
    frames.append(np.zeros_like(scene.depth))                      # nothing but background
    frames.append(np.full_like(scene.depth, 2.0))                  # a wall through the box
    fracs = []
    for k, dep in enumerate(frames):
        ref = oracle_run(orc, scene, ctx, inv, depth_override=dep, use_bricks=False)
        for skip, elide in ((False, False), (True, False), (True, True), (False, False)):
            ctx.set_skip_background(skip)
            ctx.set_elide_stores(elide)
            for _ in range(2):   # twice: the second sweep of an elided pair runs against recorded tile states
                ctx.step(dep, scene.color)
            got = ctx.readback_tsdf()
            assert same_bits(got, ref["tsdf"]), "frame %d skip %d elide %d: %d voxels differ" % (
                k, skip, elide, count_diff(got, ref["tsdf"]))
        skipped, total = ctx.skipped_pairs()
        assert total == np.prod(ctx.geo.tiles) * n
        fracs.append(skipped / total)
    assert fracs[2] >= fracs[0] > 0.05 and fracs[2] > 0.5, fracs   # mostly background / nothing but background
    ctx.close()


def test_background_skip_verdicts_follow_lut_and_limit_changes(pkg, orc):
    """the verdicts are per frame AND per LUT / truncation limit: integrate again after either changed, without
    a new process_textures in between"""
    scene, ctx, inv = build(pkg, n=2, wh=(128, 106), G=64)
    ctx.set_use_bricks(False)
    ctx.set_skip_background(True)
    ctx.step(scene.depth, scene.color)
    first = ctx.skipped_pairs()[0]
    swapped = [inv[1], inv[0]]
    for i in range(2):
        ctx.set_inverse_calibration(i, swapped[i], tuple(ctx.geo.res_volume))
    ctx.integrate()
    ref = oracle_run(orc, scene, ctx, swapped, use_bricks=False)
    assert same_bits(ctx.readback_tsdf(), ref["tsdf"])
    ctx.set_tsdf_limit(0.04)
    ctx.integrate()
    ref = oracle_run(orc, scene, ctx, swapped, use_bricks=False, limit=np.float32(0.04))
    assert same_bits(ctx.readback_tsdf(), ref["tsdf"])
    assert ctx.skipped_pairs()[0] != first
    v = ctx.readback_skip_tables(0)
    assert v.shape == (int(np.prod(ctx.geo.tiles)), 2) and set(np.unique(v)) <= {0, 1, 2, 3}
    ctx.close()


def test_background_skip_window_bounds_table(pkg):
    """rgbdr_readback_skip_tables(2): what the squares of 4 / 8 / 16 edge-clamped texels at every window origin
    in [-1, W-1] x [-1, H-1] have in common, against a sliding-window restatement over the frame's own images
    (widths that are not a multiple of the kernel's 32 origins per block, holes and non-finite depths included)"""
    from numpy.lib.stride_tricks import sliding_window_view
    for wh, holes in (((128, 106), False), ((75, 41), True), ((33, 97), True)):
        scene, ctx, inv = build(pkg, n=2, wh=wh, G=32)
        if holes:
            rng = np.random.default_rng(wh[0])
            scene.depth[:, rng.integers(0, wh[1], 40), rng.integers(0, wh[0], 40)] = 0.0
        ctx.set_use_bricks(False)
        ctx.set_skip_background(True)
        ctx.step(scene.depth, scene.color)
        got = ctx.readback_skip_tables(2)
        assert got.shape == (2, 3, 3, wh[1] + 1, wh[0] + 1)
        for s in range(2):
            d = ctx.readback_image(pkg.capi.IMG_DEPTH_B_RG, s)[..., 0]
            sil = ctx.readback_image(pkg.capi.IMG_SILHOUETTE, s)
            num, bg = ~np.isnan(d), sil < 1.0
            planes = (np.where(bg & num, d, np.inf), np.where(~bg & num, d, -np.inf), np.where(~bg & num, d, np.inf))
            for k, size in enumerate((4, 8, 16)):
                for b, a in enumerate(planes):
                    w = sliding_window_view(np.pad(a, ((1, 16), (1, 16)), mode="edge"), (size, size))[:wh[1] + 1, :wh[0] + 1]
                    want = w.min(axis=(2, 3)) if b == 1 else w.max(axis=(2, 3))
                    assert same_bits(got[s, k, b], want.astype(np.float32)), (wh, s, size, b)
        ctx.close()


@pytest.mark.parametrize("n", [1, 5, 6, 7, 8])
def test_background_skip_every_sensor_count(pkg, orc, n):
    """the classifier and the listed-tile kernel are instantiated per sensor count: 1 .. 8 sensors (2, 3, 4 above), on a
    grid whose tile count is not a multiple of the classifier's 1024 tiles per block"""
    scene, ctx, inv = build(pkg, n=n, wh=(96, 80), G=40)
    ctx.set_use_bricks(False)
    ref = oracle_run(orc, scene, ctx, inv, use_bricks=False)
    for skip in (True, False, True):
        ctx.set_skip_background(skip)
        ctx.step(scene.depth, scene.color)
        assert same_bits(ctx.readback_tsdf(), ref["tsdf"]), (n, skip)
    skipped, total = ctx.skipped_pairs()
    assert total == np.prod(ctx.geo.tiles) * n and skipped > 0
    ctx.close()


def test_background_skip_on_slabs_and_resampled_luts(pkg, orc):
    scene, ctx, inv = build(pkg, n=3, wh=(128, 106), G=64, inv_res=(40, 44, 48), slab_rank=1, slab_count=3)
    ctx.set_use_bricks(False)
    g = ctx.geo
    ref = oracle_run(orc, scene, ctx, inv, use_bricks=False)
    for skip in (False, True):
        ctx.set_skip_background(skip)
        ctx.step(scene.depth, scene.color)
        got = ctx.readback_tsdf()
        assert same_bits(got, ref["tsdf"][g.slab_voxel_z0:g.slab_voxel_z1])
    assert ctx.skipped_pairs()[0] > 0
    ctx.close()


def test_full_sweep_store_elision(pkg, orc):
    """RGBDR_FLAG_ELIDE_STORES: identical volumes, through every event that changes which
    tiles hold -limit (another frame, the brick sweep in between, another limit, settle)"""
    scene, ctx, inv = build(pkg, G=64)
    scene2 = pkg.synth.Scene(2, 128, 106, lut_res=(32, 27, 32), seed=4321, sphere_r=0.6)
    ctx.set_use_bricks(False)

    def expect(sc, **kw):
        return oracle_run(orc, sc, ctx, inv, use_bricks=False, **kw)["tsdf"]

    a, b = expect(scene), expect(scene2)
    ctx.set_elide_stores(True)
    for sc, want in ((scene, a), (scene, a), (scene2, b), (scene, a)):
        ctx.step(sc.depth, sc.color)
        assert same_bits(ctx.readback_tsdf(), want)
    ctx.set_use_bricks(True)                                    # brick sweep in between shares tile_state
    ctx.step(scene2.depth, scene2.color)
    ctx.set_use_bricks(False)
    ctx.step(scene.depth, scene.color)
    assert same_bits(ctx.readback_tsdf(), a)
    ctx.set_tsdf_limit(0.02)
    ctx.integrate()
    assert same_bits(ctx.readback_tsdf(), expect(scene, limit=np.float32(0.02)))
    ctx.set_tsdf_limit(0.01)
    ctx.settle(0.1)                                             # scribbles over the volume
    ctx.integrate()
    assert same_bits(ctx.readback_tsdf(), a)
    ctx.set_elide_stores(False)
    ctx.integrate()
    ctx.set_elide_stores(True)
    ctx.integrate()
    assert same_bits(ctx.readback_tsdf(), a)
    ctx.close()


@pytest.mark.parametrize("seed", list(range(1, 7 + int(__import__("os").environ.get("RGBDR_EXTRA_SEEDS", "0")))))
def test_random_call_sequences(pkg, orc, seed):
    run_random_sequence(pkg, orc, seed, None)


_SLABS = [(0, 2), (1, 2), (1, 3), (2, 3), (0, 3), (1, 4), (3, 4)]      # (for G = 48: six tile layers, so up to six slabs)
@pytest.mark.parametrize("seed,slab", [(11, (0, 2)), (12, (1, 2)), (13, (1, 3)), (14, (2, 3))] +
                         [(2000 + k, _SLABS[k % len(_SLABS)]) for k in range(int(__import__("os").environ.get("RGBDR_EXTRA_SEEDS", "0")) // 4)])
def test_random_call_sequences_on_a_slab(pkg, orc, seed, slab):
    run_random_sequence(pkg, orc, seed, slab)


@pytest.mark.parametrize("seed", [21, 22, 23] + [3000 + k for k in range(int(__import__("os").environ.get("RGBDR_EXTRA_SEEDS", "0")) // 8)])
def test_random_call_sequences_with_dxt1_colour_frames(pkg, orc, seed):
    """the same machine fed with DXT1 colour blocks (compress_rgb 1, the reference's yml default): the chain reads the
    blocks themselves, whichever road brought them"""
    run_random_sequence(pkg, orc, seed, None, dxt=True)


def run_random_sequence(pkg, orc, seed, slab, dxt=False):
    """state machine check: random interleavings of the setters, both sweeps, both schedules,
    store elision, background skip, inverse-LUT re-uploads, settle, two different frames and the four ways a frame can
    arrive (rgbdr_step, pageable upload, the page-locked double buffer, device-resident) -- after every frame the volume, the images
    and the brick table equal the oracle run with the settings in force"""
    rng = np.random.default_rng(seed)
    kw = dict(slab_rank=slab[0], slab_count=slab[1]) if slab else {}
    if dxt:
        kw["compress_rgb"] = 1
    scene, ctx, inv = build(pkg, wh=(64, 53), G=48 if slab else 32, lut_res=(16, 13, 16), **kw)
    z0, z1 = ctx.geo.slab_voxel_z0, ctx.geo.slab_voxel_z1
    if slab:
        for b in range(2):
            ctx.halo_staging(b)
    scenes = [scene, pkg.synth.Scene(2, 64, 53, lut_res=(16, 13, 16), seed=99, sphere_r=0.7)]
    up_color = [s_.color for s_ in scenes]             # what is uploaded: RGB8 pixels, or DXT1 blocks
    if dxt:
        class Decoded:                                 # what the oracle sees: the colours the blocks decode to
            pass
        for k, s_ in enumerate(scenes):
            blocks = np.stack([pkg.synth.encode_dxt(s_.color[i], 1) for i in range(2)])
            d_ = Decoded()
            d_.__dict__.update(s_.__dict__)
            d_.color = np.stack([orc.decode_dxt(blocks[i], 64, 53, 1) for i in range(2)])
            scenes[k], up_color[k] = d_, blocks
    state = dict(limit=np.float32(0.01), bricks=True, filt=True, proc=True, refine=True, min_voxels=10)
    cur = 0
    G = 48 if slab else 32
    inv = list(inv)
    inv_res = tuple(ctx.geo.res_volume)
    import torch
    dev_frames = [(torch.from_numpy(np.ascontiguousarray(s.depth)).cuda(), torch.from_numpy(np.ascontiguousarray(c)).cuda())
                  for s, c in zip(scenes, up_color)]
    torch.cuda.synchronize()
    for step_no in range(28):
        op = rng.integers(0, 18)
        if op == 0:
            state["bricks"] = not state["bricks"]
            ctx.set_use_bricks(state["bricks"])
        elif op == 1:
            ctx.set_elide_stores(bool(rng.integers(0, 2)))
            ctx.set_sweep_launches(int(rng.choice([1, 2, 3, 5])))      # (schedule only: the full sweep as several launches)
        elif op == 2:
            state["limit"] = np.float32(rng.choice([0.01, 0.02, 0.035]))
            ctx.set_tsdf_limit(float(state["limit"]))
        elif op == 3:
            ctx.set_pipelined(bool(rng.integers(0, 2)))
        elif op == 4:
            state["filt"] = not state["filt"]
            ctx.filter_textures(state["filt"])
        elif op == 5:
            state["proc"] = not state["proc"]
            ctx.use_processed_depths(state["proc"])
        elif op == 6:
            state["refine"] = not state["refine"]
            ctx.refine_boundary(state["refine"])
        elif op == 7:
            state["min_voxels"] = int(rng.choice([1, 10, 40]))
            ctx.set_min_voxels_per_brick(state["min_voxels"])
        elif op == 8:
            ctx.settle(0.05)
        elif op == 9 and slab:
            ctx.set_halo_staging(int(rng.integers(-1, 2)))
        elif op == 10:
            # setBrickSize: bricks of 5 / 6 / 8 / 10 voxels (not tile multiples; at G = 48 the reference's brick ->
            # voxel lists share rows between neighbouring bricks for 5 and 8); the volume and the LUTs stay
            ctx.set_brick_size(float(rng.choice([5, 6, 8, 10])) * 2.0 / G)
        elif op == 11:
            ctx.set_skip_background(bool(rng.integers(0, 2)))
        elif op == 12:
            # the two sensors trade inverse LUTs (each voxel is then projected into the other sensor's frame)
            inv[0], inv[1] = inv[1], inv[0]
            for i in range(2):
                ctx.set_inverse_calibration(i, inv[i], inv_res)
        elif op == 13 and not slab:
            # setVoxelSize: the volume, the brick table and the LUT arena are re-allocated; the inverse LUTs have to be set
            # again -- at the new grid's resolution or at one of their own (resampled into the grid layout once)
            G = int(rng.choice([32, 40, 48]))
            ctx.set_voxel_size(2.0 / G)
            ctx.set_brick_size(8 * 2.0 / G)
            res = (G, G, G) if rng.integers(0, 2) else (int(rng.integers(20, 44)), int(rng.integers(20, 44)), int(rng.integers(20, 44)))
            inv, inv_res = list(scene.inverse(res)), res
            for i in range(2):
                ctx.set_inverse_calibration(i, inv[i], res)
            z0, z1 = ctx.geo.slab_voxel_z0, ctx.geo.slab_voxel_z1
        elif op == 14 and not slab and G == 32:            # (a power of two: voxel centres hit the texels of a 1:1 LUT exactly)
            # the device computes the inverse LUT itself (CalibrationInverter on the device, at the grid's resolution); what
            # is resident then is read back for the oracle
            for i in range(2):
                ctx.compute_inverse_calibration(i, 2)
            g_ = ctx.geo
            inv = [ctx.readback_inverse_calibration(i, 0, int(g_.res_volume[2])) for i in range(2)]
            inv_res = tuple(g_.res_volume)
        else:
            cur = int(rng.integers(0, 2))
        sc = scenes[cur]
        how = int(rng.integers(0, 4))                  # the frame arrives by a different road every time
        if how == 0:
            ctx.step(sc.depth, up_color[cur])
        else:
            if how == 1:                               # pageable host buffers
                ctx.update(sc.depth, up_color[cur])
            elif how == 2:                             # the library's page-locked double buffer
                md, mc = ctx.map_frame_buffer()
                md[:] = sc.depth.view(np.uint8).reshape(-1)
                mc[:] = up_color[cur].reshape(-1)
                ctx.upload_mapped_frame()
            else:                                      # already in HBM (the buffers may go once the call is enqueued:
                dd, dc = dev_frames[cur]               # these stay alive for the whole sequence)
                ctx.update_device(dd.data_ptr(), dc.data_ptr())
            ctx.clear_occupied_bricks()
            ctx.process_textures()
            ctx.update_occupied_bricks()
            ctx.integrate()
        ref = oracle_run(orc, sc, ctx, inv, limit=state["limit"], use_bricks=state["bricks"], filter_textures=state["filt"],
                         processed=state["proc"], refine=state["refine"], min_voxels=state["min_voxels"])
        assert same_bits(ctx.readback_tsdf(), ref["tsdf"][z0:z1]), (seed, step_no, int(op), state)
        if step_no % 7 == 0:
            check_images(ctx, ref, 2)
            assert np.array_equal(ctx.readback_brick_counters(), ref["counters"])
        if step_no % 3 == 0:   # (the occupied filter may have been left to this call: it must be the frame's)
            assert np.array_equal(ctx.get_occupied()[0], ref["occupied"]), (seed, step_no, int(op), state)
    ctx.close()


@pytest.mark.parametrize("rank,count", [(0, 2), (1, 2), (1, 3)])
def test_halo_staging_holds_the_boundary_layers(pkg, rank, count):
    """rgbdr_set_halo_staging: after integrate the staging set holds the slab's first / last halo
    layers -- written by the full-sweep kernel itself, copied after the other sweeps"""
    import torch

    from rgbd_recon_amd import dist as rdist

    dev = torch.device("cuda:0")
    scene, ctx, _ = build(pkg, G=64, tsdf_limit=0.1, slab_rank=rank, slab_count=count)      # 2 halo layers
    assert ctx.geo.halo_tile_layers == 2
    with pytest.raises(pkg.capi.RgbdrError):
        ctx.set_halo_staging(0)                                   # not allocated yet
    sets = []
    for b in range(2):
        lo, hi, n = ctx.halo_staging(b)
        sets.append((rdist.wrap_device_floats(lo, n // 4, dev), rdist.wrap_device_floats(hi, n // 4, dev)))
    send_lo, send_hi, _, _ = rdist.halo_views(ctx.device_tsdf(), dev)
    assert send_lo.numel() * 4 == n
    for k, (bricks, elide) in enumerate([(False, False), (True, False), (False, True), (False, False)]):
        b = k & 1
        ctx.set_use_bricks(bricks)
        ctx.set_elide_stores(elide)
        sets[b][0].fill_(7.0)
        sets[b][1].fill_(7.0)
        torch.cuda.synchronize()
        ctx.set_halo_staging(b)
        ctx.step(scene.depth, scene.color)
        ctx.sync()
        has_lo, has_hi = rank > 0, rank < count - 1
        assert torch.equal(sets[b][0].view(torch.int32), send_lo.view(torch.int32)) == has_lo
        assert torch.equal(sets[b][1].view(torch.int32), send_hi.view(torch.int32)) == has_hi
        if not has_lo:
            assert bool((sets[b][0] == 7.0).all())                # a face without a neighbour is left alone
    ctx.set_halo_staging(-1)
    sets[0][1].fill_(7.0)
    ctx.step(scene.depth, scene.color)
    ctx.sync()
    assert bool((sets[0][1] == 7.0).all())
    whole = build(pkg, G=64)[1]
    with pytest.raises(pkg.capi.RgbdrError):
        whole.halo_staging(0)                                     # not a slab context
    whole.close()
    ctx.close()


@pytest.mark.parametrize("G,slab", [(64, None), (72, None), (64, (1, 2)), (40, None)])
def test_full_sweep_in_several_launches(pkg, orc, G, slab):
    """rgbdr_set_sweep_launches: the full sweep as n launches over consecutive tile ranges stores what one launch stores
    (the oracle's volume, bit for bit) -- with the XCD-ordered tile mapping (64: the tile count divides into rounds of 8
    x-rows) and without (72, 40), plain, with store elision, and with the halo staging of a slab filled by the kernel"""
    import torch

    from rgbd_recon_amd import dist as rdist

    kw = dict(slab_rank=slab[0], slab_count=slab[1]) if slab else {}
    scene, ctx, inv = build(pkg, n=3, G=G, tsdf_limit=0.1, **kw)
    whole = build(pkg, n=3, G=G, tsdf_limit=0.1)[1] if slab else ctx
    want = oracle_run(orc, scene, whole, inv)["tsdf"]
    if slab:
        whole.close()
    if slab:
        want = want[ctx.geo.slab_voxel_z0:ctx.geo.slab_voxel_z1]
        lo, hi, nbytes = ctx.halo_staging(0)
        stage_lo = rdist.wrap_device_floats(lo, nbytes // 4, torch.device("cuda:0"))     # (rank 1 of 2: the lower face has the neighbour)
        ctx.set_halo_staging(0)
    with pytest.raises(pkg.capi.RgbdrError):
        ctx.set_sweep_launches(0)
    with pytest.raises(pkg.capi.RgbdrError):
        ctx.set_sweep_launches(65)
    for n in (1, 2, 3, 7, 64):
        for elide in (False, True):
            ctx.set_sweep_launches(n)
            ctx.set_elide_stores(elide)
            for _ in range(2):
                ctx.step(scene.depth, scene.color)
            got = ctx.readback_tsdf()
            assert same_bits(got, want), "%d launches, elide %d: %d voxels differ" % (n, elide, count_diff(got, want))
            if slab and not elide:
                send_lo = rdist.halo_views(ctx.device_tsdf(), torch.device("cuda:0"))[0]
                ctx.sync()
                assert torch.equal(stage_lo.view(torch.int32), send_lo.view(torch.int32))
                stage_lo.fill_(7.0)
    ctx.close()


def test_occupied_filter_is_a_snapshot_whoever_evaluates_it(pkg, orc):
    """rgbdr_update_occupied_bricks may leave the filter to its first consumer (the brick sweep folds it into its
    first kernel): the result is that of the moment of the call -- later changes of the threshold or of the
    counters do not leak into it -- whichever consumer comes first, and in both schedules"""
    scene, ctx, inv = build(pkg)
    ctx.step(scene.depth, scene.color)
    ref = oracle_run(orc, scene, ctx, inv)
    want = ref["occupied"]
    for pipelined in (False, True):
        ctx.set_pipelined(pipelined)
        for first in ("integrate", "get_occupied", "process_textures", "clear", "min_voxels", "switch"):
            ctx.set_min_voxels_per_brick(10)
            ctx.clear_occupied_bricks()
            ctx.process_textures()
            ctx.update_occupied_bricks()
            if first == "min_voxels":
                ctx.set_min_voxels_per_brick(100000)     # after the update: must not matter
            elif first == "process_textures":
                ctx.process_textures()                   # counters double, the filter result stays
            elif first == "clear":
                ctx.clear_occupied_bricks()
                ctx.readback_brick_counters()            # flushes the clear
            elif first == "switch":
                ctx.set_pipelined(not pipelined)
                ctx.set_pipelined(pipelined)
            if first != "get_occupied":
                ctx.integrate()
                assert same_bits(ctx.readback_tsdf(), ref["tsdf"]), (pipelined, first)
            ids, _ = ctx.get_occupied()
            assert np.array_equal(ids, want), (pipelined, first)
        # two more frames counted (the counters alternate between two buffers) without an update: the filter result
        # is still the one of the last update
        ctx.clear_occupied_bricks()
        ctx.process_textures()
        ctx.update_occupied_bricks()
        empty = np.zeros_like(scene.depth)
        for _ in range(2):
            ctx.clear_occupied_bricks()
            ctx.update(empty, scene.color)
            ctx.process_textures()
        assert ctx.readback_brick_counters().sum() == 0
        assert np.array_equal(ctx.get_occupied()[0], want), pipelined
        ctx.update(scene.depth, scene.color)
    ctx.close()


def test_deferred_counter_clear(pkg):
    """clearOccupiedBricks is performed by the next process_textures -- or by whoever reads the
    counters first"""
    scene, ctx, _ = build(pkg)
    ctx.step(scene.depth, scene.color)
    c1 = ctx.readback_brick_counters()
    assert c1.sum() > 0
    ctx.clear_occupied_bricks()
    assert ctx.readback_brick_counters().sum() == 0            # flushed by the reader
    ctx.process_textures()
    assert np.array_equal(ctx.readback_brick_counters(), c1)
    ctx.process_textures()                                      # no clear in between: the reference double counts too
    assert np.array_equal(ctx.readback_brick_counters(), 2 * c1)
    ctx.clear_occupied_bricks()
    ctx.update_occupied_bricks()                                # flushed by the occupied update
    assert len(ctx.get_occupied()[0]) == 0
    ctx.clear_occupied_bricks()
    ctx.process_textures()
    ctx.update_occupied_bricks()
    assert np.array_equal(ctx.readback_brick_counters(), c1) and len(ctx.get_occupied()[0]) > 0
    ctx.close()


def test_colour_view_of_dxt_frames_follows_later_uploads(pkg, orc):
    """rgbdr_device_image(RGBDR_IMG_COLOR) of DXT frames: the zero-copy view is documented as rewritten by every
    upload.  The blocks are decoded lazily until a view is handed out; from then on every upload decodes as well, so
    a consumer that kept the pointer of frame A sees frame B after B's upload (it used to see A's colours)."""
    import torch
    capi, synth = pkg.capi, pkg.synth
    scene, ctx, inv = build(pkg, compress_rgb=1)
    W, H = 128, 106
    frame_a = np.stack([synth.encode_dxt(scene.color[i], 1) for i in range(2)])
    frame_b = np.stack([synth.encode_dxt(np.ascontiguousarray(scene.color[i][::-1, ::-1]), 1) for i in range(2)])
    assert not np.array_equal(frame_a, frame_b)
    ctx.update(scene.depth, frame_a)
    view = ctx.device_image(8, 1)

    class _U8:
        def __init__(self, ptr, n):
            self.__cuda_array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (int(ptr), False), "version": 2}

    dev_view = torch.as_tensor(_U8(view.ptr, W * H * 3), device="cuda:0")          # what a device consumer keeps
    ctx.sync()
    assert np.array_equal(dev_view.cpu().numpy().reshape(H, W, 3), orc.decode_dxt(frame_a[1], W, H, 1))
    ctx.update(scene.depth, frame_b)                                                 # no further rgbdr_device_image call
    ctx.sync()
    assert np.array_equal(dev_view.cpu().numpy().reshape(H, W, 3), orc.decode_dxt(frame_b[1], W, H, 1))
    # and the passes still see the new frame
    ctx.clear_occupied_bricks(); ctx.process_textures(); ctx.update_occupied_bricks(); ctx.integrate()

    class Decoded:
        pass

    s2 = Decoded()
    s2.__dict__.update(scene.__dict__)
    s2.color = np.stack([orc.decode_dxt(frame_b[i], W, H, 1) for i in range(2)])
    ref = oracle_run(orc, s2, ctx, inv)
    check_images(ctx, ref, 2)
    ctx.close()


def test_generated_inverse_luts_keep_the_file_layout_under_no_resample(pkg, orc):
    """RGBDR_FLAG_NO_RESAMPLE keeps every sensor's inverse LUT as the x-fastest RGBA32F volume, also the ones generated
    on the device (rgbdr_compute_inverse_calibration) -- next to uploaded LUTs of any resolution (they used to be
    refused as a mix of layouts).  On a power-of-two grid the per-frame lookup of a grid-resolution volume hits texel
    centres exactly, so the volume equals the one of the default (grid-layout) context bit for bit."""
    capi = pkg.capi
    G = 32
    vols = []
    for flags in (15, 15 | capi.FLAG_NO_RESAMPLE):
        scene, ctx, inv = build(pkg, G=G, inv_res=(45, 50, 45) if flags & capi.FLAG_NO_RESAMPLE else None, flags=flags)
        ctx.compute_inverse_calibration(0, 3)                # sensor 0 generated, sensor 1 keeps its uploaded LUT
        if flags & capi.FLAG_NO_RESAMPLE:
            lut1 = scene.inverse((G, G, G))[1]               # same content as the default context's sensor 1
            ctx.set_inverse_calibration(1, lut1, (G, G, G))
        gen = ctx.readback_inverse_calibration(0, 0, G)
        ctx.step(scene.depth, scene.color)
        vols.append((gen, ctx.readback_tsdf()))
        ctx.close()
    assert same_bits(vols[0][0][..., :3], vols[1][0][..., :3])
    assert same_bits(vols[0][1], vols[1][1]) and np.any(np.abs(vols[0][1]) < 0.01)


def test_halo_exchange_after_a_resize_is_refused(pkg):
    """a resize between rgbdr_halo_begin_step and rgbdr_halo_exchange_async frees the staging sets: the exchange
    returns RGBDR_ERR_STATE instead of sending from a null buffer"""
    capi = pkg.capi
    scene = pkg.synth.Scene(2, 128, 106, lut_res=(32, 27, 32))
    cfg = capi.make_config(2, (128, 106), voxel_size=2.0 / 64, brick_size=8 * 2.0 / 64, slab_rank=0, slab_count=2)
    ctx = capi.Context(cfg, 0)
    lib = capi.lib()
    assert lib.rgbdr_halo_begin_step(ctx._h) == 0
    ctx.set_voxel_size(2.0 / 32)
    fake_comm = C.c_void_p(1)
    rc = lib.rgbdr_halo_exchange_async(ctx._h, fake_comm, -1, 1)
    assert rc == capi.ERR_STATE if hasattr(capi, "ERR_STATE") else rc == -6
    # begin_step again but no integrate: still nothing to send
    assert lib.rgbdr_halo_begin_step(ctx._h) == 0
    assert lib.rgbdr_halo_exchange_async(ctx._h, fake_comm, -1, 1) == -6
    # ... also when an EARLIER integrate has run (the guard is per step, not "something was integrated once"): a sweep
    # without a staging set, then begin_step + exchange_async with no integrate in between would send stale layers
    inv = scene.inverse((32, 32, 32))
    for i in range(2):
        ctx.set_calibration(i, scene.xyz[i], scene.lut_res, scene.uv[i], scene.lut_res, (0.5, 4.5))
        ctx.set_inverse_calibration(i, inv[i], (32, 32, 32))
    ctx.step(scene.depth, scene.color)
    assert lib.rgbdr_halo_begin_step(ctx._h) == 0
    assert lib.rgbdr_halo_exchange_async(ctx._h, fake_comm, -1, 1) == -6
    assert b"no rgbdr_integrate has filled the staging set" in lib.rgbdr_last_error(ctx._h)
    ctx.close()


def test_sensor_shards_complete_each_other(pkg, orc):
    """rgbdr_set_sensor_shard: k contexts stand for the k ranks of a slab job, each runs the pre_* chain for n / k sensors
    only; copying the packed frame layers between them and summing the brick counters (what rgbdr_shard_allgather /
    dist.FrameGather do over RCCL) completes the frame: occupied bricks and volume equal the unsharded context's bit for
    bit, in both sweeps and both schedules; integrating a shard before the gather is refused."""
    import torch
    from rgbd_recon_amd import dist as rdist
    capi, synth = pkg.capi, pkg.synth
    dev = torch.device("cuda", 0)
    n, k = 4, 2
    scene, whole, inv = build(pkg, n=n)
    scene2 = synth.Scene(n, 128, 106, lut_res=(32, 27, 32), seed=77, sphere_r=0.7)
    ranks = []
    for r in range(k):
        c = capi.Context(whole.cfg, 0)
        for i in range(n):
            c.set_calibration(i, scene.xyz[i], scene.lut_res, scene.uv[i], scene.lut_res, (0.5, 4.5))
            c.set_inverse_calibration(i, inv[i], tuple(c.geo.res_volume))
        c.set_sensor_shard(r * (n // k), n // k)
        ranks.append(c)
    with pytest.raises(capi.RgbdrError) as e:
        ranks[0].set_sensor_shard(3, 2)
    assert e.value.status == capi.ERR_OUT_OF_RANGE

    def gather():
        views = [c.shard_view() for c in ranks]
        for c in ranks:
            c.sync()
        words = views[0].sensor_bytes // 4
        fr = [rdist.wrap_device_words(v.frames, words * n, dev) for v in views]
        cn = [rdist.wrap_device_words(v.counters, v.num_bricks, dev) for v in views]
        assert [(v.first, v.count) for v in views] == [(r * (n // k), n // k) for r in range(k)]
        total = sum(c.clone() for c in cn)
        for r, v in enumerate(views):
            lo, hi = v.first * words, (v.first + v.count) * words
            for q in range(k):
                if q != r:
                    fr[q][lo:hi] = fr[r][lo:hi]
        for c in cn:
            c.copy_(total)
        torch.cuda.synchronize()
        for c in ranks:
            with pytest.raises(capi.RgbdrError):
                c.integrate()                       # looking at the buffers (shard_view) does not disarm the guard ...
            c.shard_gather_done()                   # ... the host's word that its own gather is enqueued does

    for pipelined in (False, True):
        for bricks in (True, False):
            for c in ranks + [whole]:
                c.set_pipelined(pipelined)
                c.set_use_bricks(bricks)
            for sc in (scene, scene2, scene):
                whole.step(sc.depth, sc.color)
                want = whole.readback_tsdf()
                for c in ranks:
                    c.update(sc.depth, sc.color)
                    c.clear_occupied_bricks()
                    c.process_textures()
                with pytest.raises(capi.RgbdrError) as e:           # the other sensors' frames have not arrived
                    ranks[1].integrate()
                assert e.value.status == capi.ERR_STATE
                with pytest.raises(capi.RgbdrError):
                    ranks[0].update_occupied_bricks()
                gather()
                for r, c in enumerate(ranks):
                    c.update_occupied_bricks()
                    c.integrate()
                    assert np.array_equal(c.readback_brick_counters(), whole.readback_brick_counters())
                    assert np.array_equal(c.get_occupied()[0], whole.get_occupied()[0])
                    got = c.readback_tsdf()
                    assert same_bits(got, want), "rank %d, pipelined %s, bricks %s: %d voxels differ" % (r, pipelined, bricks, count_diff(got, want))
                    # its own sensors' images are the whole context's; the chain did not touch the others' since the start
                    for i in range(r * (n // k), (r + 1) * (n // k)):
                        assert same_bits(c.readback_image(7, i), whole.readback_image(7, i))
    # back to every sensor: an ordinary context again
    ranks[0].set_pipelined(False)
    ranks[0].set_sensor_shard(0, 0)
    ranks[0].step(scene2.depth, scene2.color)
    whole.set_pipelined(False)
    whole.step(scene2.depth, scene2.color)
    assert same_bits(ranks[0].readback_tsdf(), whole.readback_tsdf())
    for c in ranks + [whole]:
        c.close()
