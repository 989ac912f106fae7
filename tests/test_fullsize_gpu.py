"""Parity at the sizes BASELINE.json names (4-8 sensors of 512x424, 256^3 / 512^3 /
1024^3 grids).  The oracle is too slow for whole volumes of that size, so each case
checks (a) every pre_* image and the brick table of the full-size frame set against the
oracle, (b) bands of z rows of the TSDF against the oracle bit for bit (first, middle and
last tile layer), and (c) properties that do not depend on the size: a second integrate
reproduces the volume, the brick-skipping sweep equals the full sweep inside occupied
bricks and -limit outside, Z slabs concatenate to the whole volume, and the slab
ray-march composites to the single-volume frame."""
import os

import ctypes as C

import numpy as np
import pytest

from conftest import count_diff, same_bits

pytestmark = pytest.mark.gpu
BMIN, BMAX = (-1.0, 0.0, -1.0), (1.0, 2.0, 1.0)
W, H = 512, 424
IMG = {"morph": 1, "depth_rg": 2, "lab": 3, "depth_b": 4, "sil": 5, "normal": 6, "quality": 7}
_scenes = {}


def scene_for(pkg, n):
    if n not in _scenes:
        _scenes[n] = pkg.synth.Scene(n, W, H, lut_res=(128, 106, 128), seed=1234)
    return _scenes[n]


def make_ctx(pkg, scene, grid, **kw):
    capi = pkg.capi
    G = grid[0]
    ctx = capi.Context(capi.make_config(scene.N, (W, H), voxel_size=2.0 / G, brick_size=8 * 2.0 / G, res_override=grid, **kw), 0)
    for i in range(scene.N):
        ctx.set_calibration(i, scene.xyz[i], scene.lut_res, scene.uv[i], scene.lut_res, (0.5, 4.5))
        ctx.synth_inverse_calibration(i, scene.pinhole(i))
    return ctx


def check_rows(orc, ctx, tsdf_rows, z0, rows, limit=0.01):
    """oracle integrate on `rows` z rows starting at global row z0 (1:1 LUT rows from the device)"""
    n = ctx.cfg.num_sensors
    g = ctx.geo
    inv = [ctx.readback_inverse_calibration(i, z0, z0 + rows) for i in range(n)]
    sil = [ctx.readback_image(IMG["sil"], i) for i in range(n)]
    db = [ctx.readback_image(IMG["depth_b"], i) for i in range(n)]
    q = [ctx.readback_image(IMG["quality"], i) for i in range(n)]
    ref = orc.integrate(inv, sil, db, q, (g.res_volume[0], g.res_volume[1], rows), limit)
    assert same_bits(tsdf_rows, ref), "rows %d..%d: %d voxels differ" % (z0, z0 + rows, count_diff(tsdf_rows, ref))
    return ref


@pytest.mark.parametrize("G", [256, 512])
def test_four_sensors_single_gpu(pkg, orc, G):
    """configs[1] (256^3) and configs[2] (512^3, the benchmark workload)"""
    orc.set_threads(16)
    scene = scene_for(pkg, 4)
    ctx = make_ctx(pkg, scene, (G, G, G))
    g = ctx.geo
    ctx.step(scene.depth, scene.color)                         # reference default: brick-skipping sweep
    ref = orc.run_pipeline(scene, BMIN, BMAX, (G, G, G), None, brick_size=g.brick_size, bv=tuple(g.brick_voxels_axis),
                           res_bricks=tuple(g.res_bricks))
    for name, which in IMG.items():
        for i in range(4):
            got = ctx.readback_image(which, i)
            assert same_bits(got, ref[name][i]), "%s sensor %d: %d texels differ" % (name, i, count_diff(got, ref[name][i]))
    assert np.array_equal(ctx.readback_brick_counters(), ref["counters"])
    ids, ratio = ctx.get_occupied()
    assert np.array_equal(ids, ref["occupied"]) and 0.005 < ratio < 0.5
    bricked = ctx.readback_tsdf()
    ctx.set_use_bricks(False)
    ctx.integrate()
    full = ctx.readback_tsdf()
    ctx.integrate()
    assert same_bits(ctx.readback_tsdf(), full)                # the volume is a pure function of the frame
    mask = np.zeros(g.num_bricks, bool)
    mask[ids] = True
    m3 = mask.reshape(g.res_bricks[2], g.res_bricks[1], g.res_bricks[0])
    vox = np.repeat(np.repeat(np.repeat(m3, 8, 0), 8, 1), 8, 2)
    assert same_bits(bricked[vox], full[vox])
    assert np.all(bricked[~vox] == np.float32(-0.01))
    del bricked
    touched = 0
    for z0 in (0, G // 2 - 8, G // 2, G - 8):
        r = check_rows(orc, ctx, full[z0:z0 + 8], z0, 8)
        touched += int((np.abs(r) < np.float32(0.01)).sum())
    assert touched > 1000                                      # the bands cut through the surface
    # the same full sweep with verdicts instead of LUT planes / without re-storing constant tiles: the whole volume again
    for skip, elide in ((True, False), (False, True), (True, True)):
        ctx.set_skip_background(skip)
        ctx.set_elide_stores(elide)
        for _ in range(2):
            ctx.integrate()
        got = ctx.readback_tsdf()
        assert same_bits(got, full), "skip %d elide %d: %d voxels differ" % (skip, elide, count_diff(got, full))
    decided, pairs = ctx.skipped_pairs()
    assert pairs == (G // 8) ** 3 * 4 and decided > pairs // 2
    ctx.close()
    # Z slabs of the same grid concatenate to the whole volume
    parts = []
    for rank in range(2):
        c = make_ctx(pkg, scene, (G, G, G), slab_rank=rank, slab_count=2)
        c.set_use_bricks(False)
        c.step(scene.depth, scene.color)
        parts.append(c.readback_tsdf())
        c.close()
    assert same_bits(np.concatenate(parts, axis=0), full)


@pytest.mark.parametrize("layout", ["dense", "ring"])
def test_dense_and_moving_frames_at_the_benchmark_size(pkg, orc, layout):
    """bench.py's `scenes` inputs at BASELINE's size (4 sensors 512 x 424 -> 512^3): the DENSE scene (every pixel valid and
    inside the box) and MOVING frames (two different frames in a row, then the first again), through the brick sweep, the
    full sweep, the background skip and store elision -- images, brick table, occupied list against the oracle, TSDF bands
    against the oracle bit for bit, and the sweeps against each other on the whole volume"""
    orc.set_threads(16)
    G = 512
    scene = pkg.synth.Scene(4, W, H, lut_res=(128, 106, 128), seed=1234, layout=layout)
    if layout == "dense":
        assert (scene.depth > 0).all()
    frames = [scene, scene.at_frame(2), scene]
    ctx = make_ctx(pkg, scene, (G, G, G))
    g = ctx.geo
    occupied = []
    for k, fr in enumerate(frames):
        ctx.set_use_bricks(True)
        ctx.set_skip_background(False)
        ctx.set_elide_stores(False)
        ctx.step(fr.depth, fr.color)
        ref = orc.run_pipeline(fr, BMIN, BMAX, (G, G, G), None, brick_size=g.brick_size, bv=tuple(g.brick_voxels_axis),
                               res_bricks=tuple(g.res_bricks))
        for name, which in IMG.items():
            for i in range(4):
                got = ctx.readback_image(which, i)
                assert same_bits(got, ref[name][i]), "frame %d %s sensor %d: %d texels differ" % (k, name, i, count_diff(got, ref[name][i]))
        assert np.array_equal(ctx.readback_brick_counters(), ref["counters"])
        ids, ratio = ctx.get_occupied()
        assert np.array_equal(ids, ref["occupied"])
        occupied.append(ratio)
        bricked = ctx.readback_tsdf()
        ctx.set_use_bricks(False)
        ctx.integrate()
        full = ctx.readback_tsdf()
        mask = np.zeros(g.num_bricks, bool)
        mask[ids] = True
        vox = np.repeat(np.repeat(np.repeat(mask.reshape(g.res_bricks[2], g.res_bricks[1], g.res_bricks[0]), 8, 0), 8, 1), 8, 2)
        assert same_bits(bricked[vox], full[vox]) and np.all(bricked[~vox] == np.float32(-0.01))
        del bricked, vox
        touched = 0
        for z0 in (64, G // 2 - 8, G // 2 + 120, G - 72):
            r = check_rows(orc, ctx, full[z0:z0 + 8], z0, 8)
            touched += int((np.abs(r) < np.float32(0.01)).sum())
        assert touched > 1000
        for skip, elide in ((True, False), (True, True)):
            ctx.set_skip_background(skip)
            ctx.set_elide_stores(elide)
            ctx.integrate()
            got = ctx.readback_tsdf()
            assert same_bits(got, full), "frame %d skip %d elide %d: %d voxels differ" % (k, skip, elide, count_diff(got, full))
        del full
    if layout == "dense":
        assert min(occupied) > 0.05                      # more than twice the bricks of SURVEY 8(d)'s scene
    assert occupied[0] != occupied[1]                     # the frames differ in what they occupy
    ctx.close()


def test_eight_sensors_slab_of_512(pkg, orc):
    """configs[3]: 8 sensors, 512^3 split into 4 Z slabs -- one rank's slab"""
    orc.set_threads(16)
    scene = scene_for(pkg, 8)
    ctx = make_ctx(pkg, scene, (512, 512, 512), slab_rank=2, slab_count=4)
    g = ctx.geo
    assert (g.slab_voxel_z0, g.slab_voxel_z1) == (256, 384)
    ctx.set_use_bricks(False)
    ctx.step(scene.depth, scene.color)
    slab = ctx.readback_tsdf()
    for z0 in (256, 320, 376):
        check_rows(orc, ctx, slab[z0 - 256:z0 - 256 + 8], z0, 8)
    assert (np.abs(slab) < np.float32(0.01)).mean() > 1e-4
    ctx.close()


def frames_equal(a, b):
    return all(same_bits(x, y) for x, y in zip(a, b))


def test_eight_sensors_1024_slabs_with_post_pass(pkg, orc):
    """configs[4]: 8 sensors, 1024^3 in 8 Z slabs, ray-marched across the slabs, then the
    inpaint / colorfill post-pass.  The eight slab contexts (13.4 GB each) and the
    single-context volume they are compared with share the one GPU of the test box."""
    import torch

    from rgbd_recon_amd import dist as rdist

    orc.set_threads(16)
    dev = torch.device("cuda:0")
    free, _ = torch.cuda.mem_get_info()
    if free < 232e9:
        pytest.skip("needs 232 GB of free HBM for the 1024^3 volume next to its eight slabs")
    scene = scene_for(pkg, 8)
    grid = (1024, 1024, 1024)
    view = pkg.capi.make_view((2.2, 1.6, 1.9), (0.0, 0.9, 0.0), (0.0, 1.0, 0.0), 50.0, 640, 360, BMIN, BMAX, shade_mode=0)
    os.environ["RGBDR_ARENA_TRIALS"] = "1"                    # no spare HBM for candidate placements here
    try:
        run_1024(pkg, orc, torch, rdist, dev, scene, grid, view)
    except pkg.capi.RgbdrError as e:
        if "memory" in str(e).lower():
            pytest.skip("HBM ran out next to another tenant: %s" % e)
        raise
    finally:
        os.environ.pop("RGBDR_ARENA_TRIALS", None)


def run_1024(pkg, orc, torch, rdist, dev, scene, grid, view):
    live = []
    try:
        run_1024_body(pkg, orc, torch, rdist, dev, scene, grid, view, live)
    finally:
        for c in live:
            c.close()


def run_1024_body(pkg, orc, torch, rdist, dev, scene, grid, view, live):
    whole = make_ctx(pkg, scene, grid)
    live.append(whole)
    whole.step(scene.depth, scene.color)                       # bricks on: peels available
    whole.set_use_bricks(False)
    whole.integrate()
    want = {}
    for skip in (0, 1):
        view.skip_space = skip
        want[skip] = whole.raymarch(view)
        want[skip, "fill"] = whole.fill_colors(view.width, view.height)
    assert 0.05 < (want[0][1] < 1).mean() < 0.95
    # a band of the 1024^3 volume against the oracle (one rank's first tile layer)
    ctxs = []
    for rank in range(8):
        c = make_ctx(pkg, scene, grid, slab_rank=rank, slab_count=8)
        live.append(c)
        c.step(scene.depth, scene.color)
        c.set_use_bricks(False)
        c.integrate()
        c.sync()
        ctxs.append(c)
    g3 = ctxs[3].geo
    assert (g3.slab_voxel_z0, g3.slab_voxel_z1) == (384, 512)
    slab = ctxs[3].readback_tsdf()
    check_rows(orc, ctxs[3], slab[120:124], 384 + 120, 4)
    del slab
    views = [rdist.halo_views(c.device_tsdf(), dev) for c in ctxs]
    for r in range(7):
        views[r + 1][2].copy_(views[r][1])
        views[r][3].copy_(views[r + 1][0])
    torch.cuda.synchronize()
    npix = view.width * view.height
    for skip in (0, 1):
        view.skip_space = skip
        ks = [rdist.wrap_device_int32(c.raymarch_find(view), npix, dev) for c in ctxs]
        kmin = torch.stack(ks).min(dim=0).values
        for k in ks:
            k.copy_(kmin)
        torch.cuda.synchronize()
        color = np.tile(np.float32([0, 1, 0, 0]), (view.height, view.width, 1))
        depth = np.ones((view.height, view.width), np.float32)
        owners = 0
        for c, k in zip(ctxs, ks):
            cc, dd, nn = c.raymarch_shade(view)
            mine = (k.cpu().numpy() != rdist.NO_HIT).reshape(view.height, view.width)
            color[mine], depth[mine] = cc[mine], dd[mine]
            owners += int(mine.any())
            assert same_bits(nn, want[skip][2])
        assert owners >= 3
        assert same_bits(color, want[skip][0]) and same_bits(depth, want[skip][1])
        ctxs[0].upload_view_frame(color, depth)
        assert frames_equal(ctxs[0].fill_colors(view.width, view.height), want[skip, "fill"])


def test_a_volume_of_two_to_the_32_voxels(pkg, orc, monkeypatch):
    """2048 x 2048 x 1024 voxels from one sensor: 17 GB of TSDF and a 52 GB LUT arena, sizes a 288 GB device is built for and
    where every voxel, byte and tile index computed in 32 bits would wrap (2^32 voxels, 8.4 M tiles).  Tile layers at the
    start, either side of voxel 2^31, and at the very end equal the oracle's z rows bit for bit."""
    import torch
    free, _ = torch.cuda.mem_get_info()
    if free < 100 << 30:
        pytest.skip("needs 100 GB of free HBM")
    monkeypatch.setenv("RGBDR_ARENA_TRIALS", "1")          # one 52 GB arena, no placement shopping
    orc.set_threads(16)
    scene = scene_for(pkg, 1)
    grid = (2048, 2048, 1024)
    ctx = make_ctx(pkg, scene, grid)
    g = ctx.geo
    assert tuple(g.tiles) == (256, 256, 128) and g.tiles[0] * g.tiles[1] * g.tiles[2] * 512 == 1 << 32
    ctx.set_use_bricks(False)
    ctx.step(scene.depth, scene.color)
    ctx.sync()
    layer = np.empty(2048 * 2048 * 8, np.float32)
    seen_band, full = 0, {}
    for tz in (0, 63, 64, 127):
        ctx._chk(pkg.capi.lib().rgbdr_readback_tile_layers(ctx._h, tz, 1, layer.ctypes.data_as(C.POINTER(C.c_float))))
        rows = layer.reshape(256, 256, 8, 8, 8).transpose(2, 0, 3, 1, 4).reshape(8, 2048, 2048)     # [ty, tx, z, y, x] -> [z, Y, X]
        ref = check_rows(orc, ctx, rows, tz * 8, 8)
        seen_band += int((np.abs(ref) < np.float32(0.01)).sum())
        full[tz] = layer.copy()
    assert seen_band > 1000                                  # the surface crosses the layers looked at
    # the brick-skipping sweep over the same 8.4 M tiles (work list, tile states, clears): the full sweep's values inside
    # occupied bricks, -limit everywhere else.  The z voxels are twice as long as the others on this grid, so a brick is
    # 8 x 8 x 4 voxels: 16.8 M bricks, two to a tile; one pixel marks a brick (a brick is smaller than a pixel's footprint)
    ctx.set_min_voxels_per_brick(1)
    ctx.set_use_bricks(True)
    ctx.step(scene.depth, scene.color)
    ctx.sync()
    occupied = np.zeros(g.num_bricks, bool)
    occupied[ctx.get_occupied()[0]] = True
    assert 0 < occupied.sum() < occupied.size and tuple(g.res_bricks) == (256, 256, 256) and tuple(g.brick_voxels_axis) == (8, 8, 4)
    drawn = 0
    for tz in (0, 63, 64, 127):
        ctx._chk(pkg.capi.lib().rgbdr_readback_tile_layers(ctx._h, tz, 1, layer.ctypes.data_as(C.POINTER(C.c_float))))
        pair = occupied.reshape(256, 256 * 256)[2 * tz:2 * tz + 2]                                    # the two brick layers of this tile layer
        occ = np.repeat(pair.T[:, :, None], 256, axis=2).reshape(-1)                                  # [tile][z half][256 voxels]
        assert same_bits(layer[occ], full[tz][occ]) and np.all(layer[~occ] == np.float32(-0.01)), tz
        drawn += int(occ.sum())
    assert drawn > 0
    ctx.close()


def test_a_volume_that_does_not_fit_is_refused_and_leaves_the_device_usable(pkg, orc, monkeypatch):
    """a voxel size too fine for the device (4096 x 4096 x 2048 voxels: 137 GB of TSDF, 412 GB of LUT planes per sensor): whichever
    allocation fails, the call returns a HIP error status -- no crash, nothing left behind -- and the next context of an
    ordinary size reproduces the oracle"""
    import torch
    monkeypatch.setenv("RGBDR_ARENA_TRIALS", "1")
    capi = pkg.capi
    scene = scene_for(pkg, 1)
    warm = make_ctx(pkg, scene, (64, 64, 64))                # (the runtime's first-use allocations: code objects, pools)
    warm.step(scene.depth, scene.color)
    warm.close()
    torch.cuda.synchronize()
    free0, total = torch.cuda.mem_get_info()
    cfg = capi.make_config(1, (W, H), voxel_size=2.0 / 4096, brick_size=8 * 2.0 / 4096, res_override=(4096, 4096, 2048))
    refused = False
    try:
        ctx = capi.Context(cfg, 0)                           # the TSDF alone may or may not fit
    except capi.RgbdrError as e:
        refused = True
        assert e.status == capi.ERR_HIP, e
    else:
        ctx.set_calibration(0, scene.xyz[0], scene.lut_res, scene.uv[0], scene.lut_res, (0.5, 4.5))
        with pytest.raises(capi.RgbdrError) as e:            # 412 GB of LUT planes never do
            ctx.synth_inverse_calibration(0, scene.pinhole(0))
        assert e.value.status in (capi.ERR_HIP, capi.ERR_STATE), e.value
        refused = True
        ctx.close()
    assert refused
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 64 << 20, "%.1f MiB not returned" % ((free0 - free1) / 2 ** 20)
    small = make_ctx(pkg, scene, (64, 64, 64))
    small.set_use_bricks(False)
    small.step(scene.depth, scene.color)
    check_rows(orc, small, small.readback_tsdf(), 0, 64)
    small.close()


def test_one_recorded_stream_128(pkg, orc, tmp_path):
    """configs[0]: one sensor's `.stream` recording (512x424 f32 depth + RGB8 colour in the
    reference's frame layout) and LUT files through the C++ host mirror into a 128^3 TSDF,
    the whole volume against the oracle"""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "rgbd-recon_amd", "host", "frame_loop")
    G = 128
    scene = pkg.synth.Scene(1, W, H, lut_res=(128, 106, 128), seed=1234)
    inv = scene.inverse((G, G, G))
    d = str(tmp_path)
    os.makedirs(os.path.join(d, "recordings"))
    assert orc.lut_write(os.path.join(d, "s0.cv_xyz"), scene.xyz[0], 3) == 0
    assert orc.lut_write(os.path.join(d, "s0.cv_uv"), scene.uv[0], 2) == 0
    assert orc.lut_write(os.path.join(d, "s0.cv_xyz_inv"), inv[0], 4) == 0
    with open(os.path.join(d, "recordings", "s0.stream"), "wb") as f:
        f.write(scene.color[0].tobytes())
        f.write(scene.depth[0].tobytes())
    out = os.path.join(d, "out.tsdf")
    r = subprocess.run([exe, d, "1", str(W), str(H), str(G), out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = np.fromfile(out, dtype=np.float32).reshape(G, G, G)
    g = pkg.capi.compute_geometry(pkg.capi.make_config(1, (W, H), voxel_size=2.0 / G, brick_size=8 * 2.0 / G))
    orc.set_threads(16)
    ref = orc.run_pipeline(scene, BMIN, BMAX, (G, G, G), inv, brick_size=g.brick_size, bv=g.brick_voxels,
                           res_bricks=tuple(g.res_bricks))
    assert same_bits(got, ref["tsdf"]), "%d voxels differ" % count_diff(got, ref["tsdf"])
    assert (np.abs(got) < np.float32(0.01)).sum() > 1000
