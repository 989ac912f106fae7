"""Brick -> voxel membership: the product's tables (csrc/geometry.cpp compute_brick_tables, read through
rgbdr_brick_voxel_range) against the reference's own construction -- divideBox
(framework/reconstruction/recon_integration.cpp:361-388) + VolumeSampler::containedVoxels
(framework/rendering/volume_sampler.cpp:50-62) -- restated twice: literally in the C oracle
(orc_brick_voxel_mask) and, per axis, in numpy float32 below.  CPU only."""
import numpy as np
import pytest

F = np.float32

DEFAULT = dict(bbox_min=(-1.0, 0.0, -1.0), bbox_max=(1.0, 2.2, 1.0), voxel_size=0.01, brick_size=0.1)
# last brick of x and of y reaches one index past the axis end (linear-index aliasing, see voxel_occupied)
OVERFLOW = dict(bbox_min=(-1.0, 0.0, -1.0), bbox_max=(1.4, 2.4, 1.0), voxel_size=0.02, brick_size=0.1)
CONFIGS = {
    "reference default 200x221x200, 10-voxel bricks": DEFAULT,
    "benchmark 512^3, 8-voxel bricks": dict(bbox_min=(-1.0, 0.0, -1.0), bbox_max=(1.0, 2.0, 1.0), voxel_size=2.0 / 512,
                                            brick_size=8 * 2.0 / 512),
    "non-divisible 0.013 / 0.05": dict(bbox_min=(-1.0, 0.0, -1.0), bbox_max=(0.5, 1.3, 2.0), voxel_size=0.013, brick_size=0.05),
    "calib_inverter spacing 0.007 / 0.1": dict(bbox_min=(-1.0, 0.0, -1.0), bbox_max=(1.0, 2.2, 1.0), voxel_size=0.007,
                                               brick_size=0.1),
    "x and y overflow": OVERFLOW,
}


def axis_ranges_f32(mn, mx, brick_size, dim):
    """containedVoxels' loop bounds of every brick of one axis, in binary32 like glm::fvec3"""
    mn, mx, bs = F(mn), F(mx), F(brick_size)
    size, start, step = F(mx - mn), mn, F(F(1.0) / F(dim))
    out = []
    while F(F(size - start) + mn) > 0:
        bsz = min(bs, F(F(size - start) + mn))
        pos_n, size_n = F(F(start - mn) / size), F(bsz / size)
        lo = int(F(pos_n / step))                       # float -> unsigned truncation
        bound = F(F(pos_n + size_n) / step)
        hi, x = lo - 1, lo
        while F(x) < bound:                             # `x < (pos + size) / step`, x converted to float
            hi, x = x, x + 1
        out.append((lo, hi))
        start = F(start + bs)
    return out


def product_ranges(capi, cfg):
    g = capi.compute_geometry(cfg)
    return g, [[capi.brick_voxel_range(cfg, a, b) for b in range(g.res_bricks[a])] for a in range(3)]


def mask_from_ranges(ranges, res, occupied):
    """The voxels an occupied-brick mask draws according to per-axis ranges: separable OR, plus the
    aliasing of indices past the x / y end through z*X*Y + y*X + x (what voxel_occupied evaluates
    per voxel on the device, here by scattering linear indices)."""
    X, Y, Z = res
    rb = [len(r) for r in ranges]
    occ = np.asarray(occupied, bool).reshape(rb[2], rb[1], rb[0])
    out = np.zeros(X * Y * Z, bool)
    for bz, by, bx in zip(*np.nonzero(occ)):
        xs = np.arange(ranges[0][bx][0], ranges[0][bx][1] + 1)
        ys = np.arange(ranges[1][by][0], ranges[1][by][1] + 1)
        zs = np.arange(ranges[2][bz][0], ranges[2][bz][1] + 1)
        ids = (zs[:, None, None] * (X * Y) + ys[None, :, None] * X + xs[None, None, :]).ravel()
        out[ids[ids < X * Y * Z]] = True
    return out.reshape(Z, Y, X)


@pytest.mark.parametrize("name", list(CONFIGS))
def test_per_axis_ranges_match_the_float32_restatement(pkg, orc, name):
    capi = pkg.capi
    kw = CONFIGS[name]
    cfg = capi.make_config(1, (16, 16), **kw)
    g, ranges = product_ranges(capi, cfg)
    assert tuple(g.res_bricks) == orc.divide_box(kw["bbox_min"], kw["bbox_max"], g.brick_size)
    for a in range(3):
        want = axis_ranges_f32(kw["bbox_min"][a], kw["bbox_max"][a], g.brick_size, g.res_volume[a])
        assert ranges[a] == want, "axis %d" % a


def test_reference_default_grid_known_answers(pkg, orc):
    """The reference's own operating point (kinect_client.cpp:87-93): 200 x 221 x 200 voxels, 0.1 m bricks."""
    capi = pkg.capi
    cfg = capi.make_config(1, (16, 16), **DEFAULT)
    g, r = product_ranges(capi, cfg)
    assert tuple(g.res_volume) == (200, 221, 200)
    assert tuple(g.res_bricks) == (20, 22, 20) and g.num_bricks == 8800     # not ceil(221 / 10) = 23 on y
    # brick_size is 0.1f rounded down (0.099999994): the upper bound (pos + size) / step of most inner bricks
    # rounds to just above the next multiple of ten, so the brick also lists the next brick's first row
    assert r[0][0] == (0, 9) and r[0][1] == (10, 20) and r[0][2] == (20, 30) and r[0][19] == (190, 199)
    # 221 rows over 22 bricks of 10.045 rows: every brick also holds the first row of the next one
    assert r[1][0] == (0, 10) and r[1][1] == (10, 20) and r[1][20] == (200, 210) and r[1][21] == (210, 220)
    assert r[2] == r[0]
    # every voxel is listed by some brick, none by more than two per axis
    for a in range(3):
        cover = np.zeros(g.res_volume[a], int)
        for lo, hi in r[a]:
            cover[lo:hi + 1] += 1
        assert cover.min() == 1 and cover.max() == 2
    # how far this is from an integer partition voxel // 10 (what round 1 built): rows shared by two bricks
    shared = [sum(1 for v in range(g.res_volume[a]) if sum(lo <= v <= hi for lo, hi in r[a]) == 2) for a in range(3)]
    assert shared == [17, 21, 17]      # x/z: bricks 0, 15 and 19 end on their own last row
    frac = 1.0 - np.prod([1.0 - s / n for s, n in zip(shared, g.res_volume)])
    assert 0.24 < frac < 0.26          # a quarter of the voxels are listed by more than one brick


@pytest.mark.parametrize("name", ["reference default 200x221x200, 10-voxel bricks", "non-divisible 0.013 / 0.05",
                                  "x and y overflow"])
def test_separable_tables_reproduce_the_reference_index_lists(pkg, orc, name):
    """orc_brick_voxel_mask runs the reference's nested loops literally (index lists, linear ids);
    the product's per-axis ranges + the aliasing rule must draw exactly the same voxels."""
    capi = pkg.capi
    kw = CONFIGS[name]
    cfg = capi.make_config(1, (16, 16), **kw)
    g, ranges = product_ranges(capi, cfg)
    res = tuple(g.res_volume)
    rng = np.random.default_rng(5)
    for density in (0.02, 0.3, 1.0):
        occ = (rng.random(g.num_bricks) < density).astype(np.uint8)
        occ[-1] = 1                                            # the corner brick: overflow on every axis that has one
        lit, rb, outside = orc.brick_voxel_mask(kw["bbox_min"], kw["bbox_max"], g.brick_size, res, occ)
        assert rb == tuple(g.res_bricks)
        mine = mask_from_ranges(ranges, res, occ)
        assert np.array_equal(lit.astype(bool), mine), "%d voxels differ" % int(np.sum(lit.astype(bool) != mine))
        if name == "x and y overflow":
            assert outside > 0
        else:
            assert outside == 0


def test_overflow_config_really_overflows(pkg):
    capi = pkg.capi
    cfg = capi.make_config(1, (16, 16), **OVERFLOW)
    g, r = product_ranges(capi, cfg)
    assert tuple(g.res_volume) == (121, 121, 100)
    assert r[0][-1][1] == 121 and r[1][-1][1] == 121 and r[2][-1][1] == 99


def test_power_of_two_grids_are_an_exact_partition(pkg):
    """every benchmark grid: bricks of 8 voxels with nothing shared, so brick mode == voxel // 8"""
    capi = pkg.capi
    for G in (64, 256, 512, 1024):
        cfg = capi.make_config(1, (16, 16), voxel_size=2.0 / G, brick_size=8 * 2.0 / G)
        g, r = product_ranges(capi, cfg)
        for a in range(3):
            assert r[a] == [(8 * b, 8 * b + 7) for b in range(G // 8)]


def test_brick_range_argument_checks(pkg):
    capi = pkg.capi
    cfg = capi.make_config(1, (16, 16), **DEFAULT)
    with pytest.raises(capi.RgbdrError) as e:
        capi.brick_voxel_range(cfg, 1, 22)
    assert e.value.status == capi.ERR_OUT_OF_RANGE
    with pytest.raises(capi.RgbdrError):
        capi.brick_voxel_range(cfg, 3, 0)
