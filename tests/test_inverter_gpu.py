"""Inverse calibration volumes generated on the device against the
oracle's exact brute-force restatement of CalibrationInverter: the device search
certifies every window it accepts (kernels_invert.hip), so every voxel must equal
the exact search bit for bit -- no fraction, no tolerance.  (The reference's CGAL
k-d tree is not available: its order among EQUIDISTANT samples is the one thing
left unpinned; the oracle and the device break ties by sample index.)"""
import numpy as np
import pytest

from conftest import same_bits

pytestmark = pytest.mark.gpu
BMIN, BMAX = (-1.0, 0.0, -1.0), (1.0, 2.0, 1.0)


def make(pkg, n=2, wh=(64, 53), G=32, lut_res=(16, 13, 16)):
    capi, synth = pkg.capi, pkg.synth
    scene = synth.Scene(n, wh[0], wh[1], lut_res=lut_res)
    ctx = capi.Context(capi.make_config(n, wh, voxel_size=2.0 / G, brick_size=8 * 2.0 / G), 0)
    for i in range(n):
        ctx.set_calibration(i, scene.xyz[i], lut_res, scene.uv[i], lut_res, (0.5, 4.5))
    return scene, ctx


@pytest.mark.parametrize("window", [0, 1, 3])
@pytest.mark.parametrize("res,lut_res", [((32, 32, 32), (16, 13, 16)), ((40, 28, 36), (24, 20, 24)), ((20, 20, 20), (6, 5, 7)),
                                         ((48, 48, 48), (40, 33, 40))])
def test_generate_inverse_lut_equals_the_exact_search(pkg, orc, res, lut_res, window):
    """whatever the first window radius: every record identical to the exhaustive search"""
    scene, ctx = make(pkg, lut_res=lut_res)
    for i in range(2):
        got = ctx.generate_inverse_lut(i, res, window=window)
        ref = orc.inverse_volume(scene.xyz[i], BMIN, BMAX, res)
        assert 0.5 < (ref[..., 3] > 0).mean() <= 1.0
        assert np.array_equal(got, ref)
        widened, exhaustive = ctx.inverse_search_stats(i)
        assert widened <= (ref[..., 3] > 0).sum() and exhaustive <= (ref[..., 3] > 0).sum()
    ctx.close()


def test_an_irregular_lattice_falls_back_to_the_exhaustive_scan_and_stays_exact(pkg, orc):
    """a forward LUT whose samples are jittered by more than a cell (the lattice folds: no window can be certified for most
    voxels) -- the device gives up the local search voxel by voxel, scans the whole volume for those, and still returns the
    exact answer; the statistics say how many took that road"""
    capi, synth = pkg.capi, pkg.synth
    lut_res = (10, 9, 11)
    scene = synth.Scene(1, 64, 53, lut_res=lut_res)
    rng = np.random.default_rng(5)
    xyz = scene.xyz[0].copy()
    interior = xyz[1:-1, 1:-1, 1:-1]
    interior += rng.normal(0.0, 0.25, interior.shape).astype(np.float32)     # the corners (frustum planes) stay
    ctx = capi.Context(capi.make_config(1, (64, 53), voxel_size=2.0 / 32, brick_size=8 * 2.0 / 32), 0)
    ctx.set_calibration(0, xyz, lut_res, scene.uv[0], lut_res, (0.5, 4.5))
    res = (24, 24, 24)
    got = ctx.generate_inverse_lut(0, res)
    ref = orc.inverse_volume(xyz, BMIN, BMAX, res)
    assert np.array_equal(got, ref)
    widened, exhaustive = ctx.inverse_search_stats(0)
    assert exhaustive > 0
    ctx.close()


@pytest.mark.parametrize("k", [0.45, -0.12, -0.3])
def test_a_radially_distorted_lattice_with_voxels_outside_the_sampled_hull_stays_exact(pkg, orc, k):
    """The certificate of kernels_invert.hip is argued for a query inside a convex, non-folding sampled region.  A lens with
    strong radial distortion bends the lattice: with k > 0 its sides cave in between the corners (pincushion), so voxels that
    pass the frustum test -- its planes come from the eight corner samples -- lie OUTSIDE the hull of the samples; with k < 0
    it bulges (k = -0.12), and at k = -0.3 the lattice FOLDS near the corners (the Jacobian 1 + 8 k + 12 k^2 of the map at (1, 1) is negative below k = -1/6) -- the library notices
    that when the calibration is set and scans exhaustively.  Every voxel must equal the exhaustive search."""
    capi, synth = pkg.capi, pkg.synth
    lut_res = (20, 17, 12)
    scene = synth.Scene(1, 64, 53, lut_res=lut_res)
    xyz = scene.xyz[0].copy()                                   # [z][y][x][3]
    nz, ny, nx = xyz.shape[:3]
    u = np.linspace(-1.0, 1.0, nx, dtype=np.float32)[None, None, :]
    v = np.linspace(-1.0, 1.0, ny, dtype=np.float32)[None, :, None]
    r2 = u * u + v * v
    scale = (1.0 + k * r2) / (1.0 + k * 2.0)                    # the four corner columns (r2 = 2) stay: the frustum planes do too
    axis = xyz[:, ny // 2:ny // 2 + 1, nx // 2:nx // 2 + 1, :]  # the central column of every depth slice
    xyz = (axis + (xyz - axis) * scale[..., None]).astype(np.float32)
    for corner in ((0, 0), (0, -1), (-1, 0), (-1, -1)):
        assert np.allclose(xyz[:, corner[0], corner[1]], scene.xyz[0][:, corner[0], corner[1]], atol=1e-5)
        xyz[:, corner[0], corner[1]] = scene.xyz[0][:, corner[0], corner[1]]       # (to the bit)
    ctx = capi.Context(capi.make_config(1, (64, 53), voxel_size=2.0 / 32, brick_size=8 * 2.0 / 32), 0)
    ctx.set_calibration(0, xyz, lut_res, scene.uv[0], lut_res, (0.5, 4.5))
    res = (36, 36, 130) if k != -0.3 else (20, 20, 66)          # (several chunks of z rows: the statistics add up over them)
    got = ctx.generate_inverse_lut(0, res)
    ref = orc.inverse_volume(xyz, BMIN, BMAX, res)
    inside = ref[..., 3] > 0
    assert 0.1 < inside.mean() < 1.0
    assert np.array_equal(got, ref), int((got != ref).any(axis=-1).sum())
    widened, exhaustive = ctx.inverse_search_stats(0)
    assert widened <= inside.sum() and exhaustive <= inside.sum()
    if k == -0.3:
        assert exhaustive == inside.sum()                       # a folded lattice: every voxel of the whole call, not of its last chunk
    else:
        assert 0 < widened and exhaustive < inside.sum()       # (a lattice this coarse and this bent certifies less than half)
    ctx.close()


@pytest.mark.parametrize("lut_res", [(1, 1, 1), (2, 2, 2), (3, 1, 2), (1, 5, 1), (8, 8, 8)])
@pytest.mark.parametrize("window", [0, 1, 8])
def test_inverter_on_calibration_volumes_with_fewer_than_eight_samples_per_axis(pkg, orc, lut_res, window):
    """cv_xyz volumes of one, two, a few cells: fewer samples than the eight neighbours the reference asks its k-d tree for,
    search windows larger than the volume.  With a window that spans the whole volume the local search IS the exhaustive
    one: frustum column identical to the exact restatement, and wherever that holds at least eight samples the same eight;
    smaller windows must stay inside the volume (values finite or the rejected voxel's -1)"""
    scene, ctx = make(pkg, lut_res=lut_res)
    res = (12, 10, 14)
    got = ctx.generate_inverse_lut(0, res, window=window)
    ref = orc.inverse_volume(scene.xyz[0], BMIN, BMAX, res)
    assert got.shape == ref.shape and np.array_equal(got[..., 3], ref[..., 3])
    assert not np.isinf(got).any()
    assert np.array_equal(got, ref)
    ctx.compute_inverse_calibration(0, window if window else 2)       # the same search at the grid's own resolution
    ctx.close()


def test_default_window_is_exact_and_reprojects(pkg, orc):
    """default first window (R = 2): the same neighbours as the exact search everywhere; the generated
    inverse composed with the forward LUT returns the voxel's world position to
    within one forward-LUT cell"""
    lut_res = (24, 20, 24)
    scene, ctx = make(pkg, lut_res=lut_res)
    res = (32, 32, 32)
    got = ctx.generate_inverse_lut(0, res)
    ref = orc.inverse_volume(scene.xyz[0], BMIN, BMAX, res)
    inside = ref[..., 3] > 0
    assert np.array_equal(got, ref)
    c = (np.arange(32) + 0.5) / 32
    Z, Y, X = np.meshgrid(c, c, c, indexing="ij")
    world = np.stack([BMIN[0] + X * 2, BMIN[1] + Y * 2, BMIN[2] + Z * 2], -1)
    idx = np.argwhere(inside)[::37]
    err = []
    for z, y, x in idx:
        u, v, d, _ = got[z, y, x]
        err.append(np.linalg.norm(orc.tex3d(scene.xyz[0], u, v, d) - world[z, y, x]))
    cell = 4.0 / lut_res[2]                                # depth spacing of the forward LUT in metres
    assert np.max(err) < cell and np.mean(err) < 0.25 * cell
    ctx.close()


def test_compute_inverse_calibration_feeds_integration(pkg, orc):
    """the on-device inverse LUT at grid resolution drives integrate(); equals running
    the oracle with the LUT the device generated, and is close to the analytic inverse"""
    scene, ctx = make(pkg, wh=(128, 106), G=64, lut_res=(32, 27, 32))
    for i in range(2):
        ctx.compute_inverse_calibration(i, 3)
    inv = [ctx.readback_inverse_calibration(i, 0, 64) for i in range(2)]
    exact = orc.inverse_volume(scene.xyz[0], BMIN, BMAX, (64, 64, 64), z_range=(24, 32))
    assert np.array_equal(inv[0][24:32, ..., :3], exact[..., :3])
    ctx.step(scene.depth, scene.color)
    g = ctx.geo
    ref = orc.run_pipeline(scene, BMIN, BMAX, (64, 64, 64), inv, brick_size=g.brick_size, bv=g.brick_voxels,
                           res_bricks=tuple(g.res_bricks))
    got = ctx.readback_tsdf()
    assert same_bits(got, ref["tsdf"])
    assert np.sum(np.abs(got) < 0.01) > 200
    ana = scene.inverse((64, 64, 64))[0]
    both = (ana[..., 3] > 0) & (inv[0][..., 0] >= 0)
    assert np.abs(ana[both][:, :3] - inv[0][both][:, :3]).mean() < 0.01
    ctx.close()
