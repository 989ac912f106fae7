"""Differential fuzz over configurations: every seed draws its own sensor count, image and colour sizes, calibration-volume
resolutions, grid (cubic, anisotropic, off the tile size), brick size, truncation limit, inverse-LUT resolution, pass
flags, sweep and schedule -- and the frame's images, brick table and volume have to equal the oracle's.  The fixed
parametrisations elsewhere pin the sizes the reference and BASELINE.json name; this one looks for the sizes nobody thought of."""
import os

import numpy as np
import pytest

from conftest import count_diff, same_bits

pytestmark = pytest.mark.gpu
BMIN, BMAX = (-1.0, 0.0, -1.0), (1.0, 2.0, 1.0)
IMG = {"morph": 1, "depth_rg": 2, "lab": 3, "depth_b": 4, "sil": 5, "normal": 6, "quality": 7}


@pytest.mark.parametrize("seed", list(range(1, 9 + int(os.environ.get("RGBDR_EXTRA_SEEDS", "0")) // 4)))
def test_random_configuration(pkg, orc, seed):
    capi, synth = pkg.capi, pkg.synth
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([1, 2, 2, 3, 4, 5, 8]))
    W, H = int(rng.integers(6, 150)), int(rng.integers(6, 120))
    color_wh = (int(rng.integers(8, 160)), int(rng.integers(8, 130))) if rng.integers(0, 3) == 0 else None
    lut_res = tuple(int(v) for v in rng.integers(2, 22, 3))
    G = int(rng.integers(9, 60))
    res = (G, G, G)
    override = (0, 0, 0)
    if rng.integers(0, 4) == 0:                            # an anisotropic grid through res_override
        res = tuple(int(v) for v in rng.integers(9, 70, 3))
        override = res
    voxel = 2.0 / res[0]
    brick = float(rng.integers(3, 13)) * voxel
    limit = float(rng.choice([0.01, 0.02, 0.05, 0.1]))
    flags = 8 | int(rng.integers(0, 8))                     # bricks bit + any of filter / processed depths / refine
    if rng.integers(0, 3):
        flags |= 7
    min_voxels = int(rng.choice([1, 5, 10, 30]))
    keep_file_layout = bool(rng.integers(0, 4) == 0)        # RGBDR_FLAG_NO_RESAMPLE: the LUT stays at its own resolution, 8-tap lookups per voxel
    if keep_file_layout:
        flags |= capi.FLAG_NO_RESAMPLE
    cfg = capi.make_config(n, (W, H), color_wh=color_wh, voxel_size=voxel, brick_size=brick, tsdf_limit=limit, min_voxels=min_voxels,
                           flags=flags, res_override=override)
    ctx = capi.Context(cfg, 0)
    g = ctx.geo
    res = tuple(g.res_volume)                              # (ceil(extent / voxel_size) in float: may be one more than G)
    scene = synth.Scene(n, W, H, lut_res=lut_res, seed=seed, color_wh=color_wh, sphere_r=float(rng.choice([0.5, 0.7, 0.9])))
    inv_res = res if rng.integers(0, 3) else tuple(int(v) for v in rng.integers(8, 50, 3))
    inv = scene.inverse(inv_res)
    for i in range(n):
        ctx.set_calibration(i, scene.xyz[i], lut_res, scene.uv[i], lut_res, (0.5, 4.5))
        ctx.set_inverse_calibration(i, inv[i], inv_res)
    desc = dict(seed=seed, n=n, wh=(W, H), color_wh=color_wh, lut_res=lut_res, res=res, inv_res=inv_res, brick_voxels=g.brick_voxels,
                limit=limit, flags=flags, min_voxels=min_voxels, keep_file_layout=keep_file_layout)
    for bricks in (bool(rng.integers(0, 2)), None):
        bricks = (not last) if bricks is None else bricks     # both sweeps, in a random order
        last = bricks
        ctx.set_use_bricks(bricks)
        ctx.set_pipelined(bool(rng.integers(0, 2)))
        ctx.set_skip_background(bool(rng.integers(0, 2)))
        ctx.set_elide_stores(bool(rng.integers(0, 2)))
        ctx.step(scene.depth, scene.color)
        ref = orc.run_pipeline(scene, BMIN, BMAX, res, inv, limit=limit, brick_size=g.brick_size, bv=g.brick_voxels,
                               res_bricks=tuple(g.res_bricks), min_voxels=min_voxels, filter_textures=bool(flags & 1),
                               processed=bool(flags & 2), refine=bool(flags & 4), use_bricks=bricks)
        for name, which in IMG.items():
            for i in range(n):
                got = ctx.readback_image(which, i)
                assert same_bits(got, ref[name][i]), (desc, name, i, count_diff(got, ref[name][i]))
        assert np.array_equal(ctx.readback_brick_counters(), ref["counters"]), desc
        assert np.array_equal(ctx.get_occupied()[0], ref["occupied"]), desc
        got = ctx.readback_tsdf()
        assert same_bits(got, ref["tsdf"]), (desc, bricks, count_diff(got, ref["tsdf"]))
    # the consumer side on the same configuration: one view of a random size, eye and shade mode through the ray-marcher
    # (with the depth peels when the last sweep left an occupied list) and the hole filling.  The ray-marcher samples the
    # LUT as it is resident -- the grid layout at the grid's resolution, which is what is read back for the oracle
    resident = inv if keep_file_layout else [ctx.readback_inverse_calibration(i, 0, res[2]) for i in range(n)]
    vw, vh = int(rng.integers(1, 90)), int(rng.integers(1, 70))
    eye = [(2.2, 1.6, 1.9), (0.85, 1.7, 0.8), (-2.0, 1.2, 2.1), (0.05, 1.95, 0.02)][int(rng.integers(0, 4))]
    view = capi.make_view(eye, (0.0, 0.9, 0.0), (0.0, 1.0, 0.0), 50.0, vw, vh, BMIN, BMAX, shade_mode=int(rng.integers(0, 4)))
    peels = None
    if last and rng.integers(0, 2):
        mask = np.zeros(g.num_bricks, np.uint8)
        mask[ctx.get_occupied()[0]] = 1
        peels = orc.depth_peels(bytes(view), BMIN, g.brick_size, tuple(g.res_bricks), ctx.readback_brick_counters(), mask)
        view.skip_space = 1
        assert same_bits(ctx.draw_depth_limits(view), peels), desc
    color, depth, ns = ctx.raymarch(view)
    db = [ctx.readback_image(IMG["depth_b"], i) for i in range(n)]
    q = [ctx.readback_image(IMG["quality"], i) for i in range(n)]
    rc, rd, rn = orc.raymarch(bytes(view), ctx.readback_tsdf(), resident, scene.uv, [scene.color[i] for i in range(n)], db, q, limit=limit, peels=peels)
    assert same_bits(ns, rn) and same_bits(depth, rd) and same_bits(color, rc), (desc, (vw, vh), view.shade_mode, view.skip_space)
    fc, fd = ctx.fill_colors(vw, vh)
    oc, od = orc.fill_colors(rc, rd)
    assert same_bits(fc, oc) and same_bits(fd, od), (desc, (vw, vh))
    ctx.close()


@pytest.mark.parametrize("seed", list(range(1, 5 + int(os.environ.get("RGBDR_EXTRA_SEEDS", "0")) // 8)))
def test_random_slab_configuration(pkg, orc, seed):
    """the same draw cut into Z slabs: k contexts (as the k ranks of a job would hold them) sweep their slabs of a random
    grid -- whole tile layers each, a halo sized by the truncation limit -- and the slabs concatenate to the oracle's volume;
    every rank counts the whole brick table"""
    capi, synth = pkg.capi, pkg.synth
    rng = np.random.default_rng(7000 + seed)
    n = int(rng.choice([1, 2, 3, 4]))
    W, H = int(rng.integers(20, 130)), int(rng.integers(20, 110))
    lut_res = tuple(int(v) for v in rng.integers(3, 20, 3))
    res = tuple(int(v) for v in rng.integers(17, 70, 3)) if rng.integers(0, 3) == 0 else (int(rng.integers(17, 64)),) * 3
    voxel = 2.0 / res[0]
    brick = float(rng.integers(3, 12)) * voxel
    limit = float(rng.choice([0.01, 0.03, 0.08]))
    override = res if res[0] != res[1] or res[1] != res[2] else (0, 0, 0)
    # (a cubic grid is ceil(extent / voxel_size) in float, setVoxelSize's rule: 2 / float32(2 / 61) is a hair above 61 -> 62 rows)
    res = tuple(capi.compute_geometry(capi.make_config(n, (W, H), voxel_size=voxel, brick_size=brick, res_override=override)).res_volume)
    tiles_z = (res[2] + 7) // 8
    halo = 1 if int(np.ceil(np.float32(limit) * np.float32(res[2]))) + 2 <= 8 else (int(np.ceil(np.float32(limit) * np.float32(res[2]))) + 2 + 7) // 8
    max_count = tiles_z // halo
    if max_count < 2:
        pytest.skip("grid too thin for two slabs with this limit")
    count = int(rng.integers(2, min(6, max_count) + 1))
    scene = synth.Scene(n, W, H, lut_res=lut_res, seed=seed, sphere_r=0.8)
    inv_res = res if rng.integers(0, 2) else tuple(int(v) for v in rng.integers(10, 48, 3))
    inv = scene.inverse(inv_res)
    bricks = bool(rng.integers(0, 2))
    parts, ref = [], None
    desc = dict(seed=seed, n=n, wh=(W, H), res=res, inv_res=inv_res, count=count, limit=limit, bricks=bricks)
    for rank in range(count):
        cfg = capi.make_config(n, (W, H), voxel_size=voxel, brick_size=brick, tsdf_limit=limit, res_override=override,
                               slab_rank=rank, slab_count=count)
        try:
            ctx = capi.Context(cfg, 0)
        except capi.RgbdrError as e:
            assert "slab" in str(e), (desc, str(e))          # (a split this grid cannot serve: refused at create, by every rank alike)
            assert rank == 0 or not parts
            pytest.skip("split refused: %s" % e)
        g = ctx.geo
        for i in range(n):
            ctx.set_calibration(i, scene.xyz[i], lut_res, scene.uv[i], lut_res, (0.5, 4.5))
            ctx.set_inverse_calibration(i, inv[i], inv_res)
        ctx.set_use_bricks(bricks)
        ctx.set_skip_background(bool(rng.integers(0, 2)))
        ctx.step(scene.depth, scene.color)
        if ref is None:
            ref = orc.run_pipeline(scene, BMIN, BMAX, res, inv, limit=limit, brick_size=g.brick_size, bv=g.brick_voxels,
                                   res_bricks=tuple(g.res_bricks), use_bricks=bricks)
        assert np.array_equal(ctx.readback_brick_counters(), ref["counters"]), (desc, rank)
        part = ctx.readback_tsdf()
        assert part.shape[0] == g.slab_voxel_z1 - g.slab_voxel_z0
        parts.append(part)
        ctx.close()
    whole = np.concatenate(parts, axis=0)
    assert same_bits(whole, ref["tsdf"]), (desc, count_diff(whole, ref["tsdf"]))
