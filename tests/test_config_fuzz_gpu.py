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
    # input formats (f-1): DXT1 / DXT5 colour blocks (compress_rgb; 1 is the reference's yml default), 8-bit depth (compress_depth;
    # coherent only without the morph-processed depths, SURVEY A.5: the processed-depths bit is cleared)
    dxt = int(rng.choice([0, 0, 1, 5]))
    u8 = bool(rng.integers(0, 6) == 0)
    if u8:
        flags &= ~2
    cfg = capi.make_config(n, (W, H), color_wh=color_wh, voxel_size=voxel, brick_size=brick, tsdf_limit=limit, min_voxels=min_voxels,
                           flags=flags, res_override=override, compress_rgb=dxt, compress_depth=int(u8))
    ctx = capi.Context(cfg, 0)
    g = ctx.geo
    res = tuple(g.res_volume)                              # (ceil(extent / voxel_size) in float: may be one more than G)
    sphere_r = float(rng.choice([0.5, 0.7, 0.9]))
    layout = "dense" if seed % 3 == 0 else "ring"          # every third configuration: every pixel valid and inside the box
    scene = synth.Scene(n, W, H, lut_res=lut_res, seed=seed, color_wh=color_wh, sphere_r=sphere_r, layout=layout)
    inv_res = res if rng.integers(0, 3) else tuple(int(v) for v in rng.integers(8, 50, 3))
    inv = scene.inverse(inv_res)
    for i in range(n):
        ctx.set_calibration(i, scene.xyz[i], lut_res, scene.uv[i], lut_res, (0.5, 4.5))
        ctx.set_inverse_calibration(i, inv[i], inv_res)
    up_depth, up_color, oracle_kw = scene.depth, scene.color, {}
    if dxt:
        Hc, Wc = scene.color.shape[1:3]
        up_color = np.stack([synth.encode_dxt(scene.color[i], dxt) for i in range(n)])

        class Decoded:                                      # the oracle sees the colours the blocks decode to
            pass
        d_ = Decoded()
        d_.__dict__.update(scene.__dict__)
        d_.color = np.stack([orc.decode_dxt(up_color[i], Wc, Hc, dxt) for i in range(n)])
        scene = d_
    if u8:
        up_depth = synth.compress_depth_u8(scene.depth)
        oracle_kw = dict(compress=True, depth_override=(up_depth.astype(np.float32) / np.float32(255.0)).astype(np.float32))
    desc = dict(seed=seed, n=n, wh=(W, H), color_wh=color_wh, lut_res=lut_res, res=res, inv_res=inv_res, brick_voxels=g.brick_voxels,
                limit=limit, flags=flags, min_voxels=min_voxels, keep_file_layout=keep_file_layout, dxt=dxt, u8=u8)
    for bricks in (bool(rng.integers(0, 2)), None):
        bricks = (not last) if bricks is None else bricks     # both sweeps, in a random order
        last = bricks
        ctx.set_use_bricks(bricks)
        ctx.set_pipelined(bool(rng.integers(0, 2)))
        ctx.set_skip_background(bool(rng.integers(0, 2)))
        ctx.set_elide_stores(bool(rng.integers(0, 2)))
        ctx.step(up_depth, up_color)
        ref = orc.run_pipeline(scene, BMIN, BMAX, res, inv, limit=limit, brick_size=g.brick_size, bv=g.brick_voxels,
                               res_bricks=tuple(g.res_bricks), min_voxels=min_voxels, filter_textures=bool(flags & 1),
                               processed=bool(flags & 2), refine=bool(flags & 4), use_bricks=bricks, **oracle_kw)
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


@pytest.mark.parametrize("seed", list(range(1, 4 + int(os.environ.get("RGBDR_EXTRA_SEEDS", "0")) // 16)))
def test_random_sensor_shard_configuration(pkg, orc, seed):
    """the pre_* chain sharded by sensor over k contexts (as the k ranks of a slab job hold them: rgbdr_set_sensor_shard), the
    packed frame layers copied between them and the brick counters summed the way rgbdr_shard_allgather / dist.FrameGather
    do it over RCCL -- random sensor counts and shard sizes, image sizes, grids, sweeps, schedules and upload roads; every
    rank's brick table, occupied list and volume equal the oracle's for three frames in a row"""
    import torch
    from rgbd_recon_amd import dist as rdist
    capi, synth = pkg.capi, pkg.synth
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(11000 + seed)
    n, k = [(2, 2), (4, 2), (4, 4), (6, 2), (6, 3), (8, 2), (8, 4), (3, 3)][int(rng.integers(0, 8))]
    W, H = int(rng.integers(20, 120)), int(rng.integers(20, 100))
    lut_res = tuple(int(v) for v in rng.integers(3, 18, 3))
    G = int(rng.integers(12, 56))
    limit = float(rng.choice([0.01, 0.03]))
    cfg = capi.make_config(n, (W, H), voxel_size=2.0 / G, brick_size=float(rng.integers(4, 11)) * 2.0 / G, tsdf_limit=limit, min_voxels=int(rng.choice([1, 10])))
    res = tuple(capi.compute_geometry(cfg).res_volume)
    scenes = [synth.Scene(n, W, H, lut_res=lut_res, seed=seed, sphere_r=0.8), synth.Scene(n, W, H, lut_res=lut_res, seed=seed + 500, sphere_r=0.6)]
    inv_res = res if rng.integers(0, 2) else tuple(int(v) for v in rng.integers(10, 40, 3))
    inv = scenes[0].inverse(inv_res)
    ranks = []
    for r in range(k):
        c = capi.Context(cfg, 0)
        for i in range(n):
            c.set_calibration(i, scenes[0].xyz[i], lut_res, scenes[0].uv[i], lut_res, (0.5, 4.5))
            c.set_inverse_calibration(i, inv[i], inv_res)
        c.set_sensor_shard(r * (n // k), n // k)
        ranks.append(c)
    g = ranks[0].geo
    desc = dict(seed=seed, n=n, k=k, wh=(W, H), res=res, inv_res=inv_res, limit=limit)
    for frame_no in range(3):
        sc = scenes[int(rng.integers(0, 2))]
        bricks, pipelined, skip = bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        for c in ranks:
            c.set_use_bricks(bricks)
            c.set_pipelined(pipelined)
            c.set_skip_background(skip)
            if rng.integers(0, 2):
                c.update(sc.depth, sc.color)
            else:
                md, mc = c.map_frame_buffer()
                md[:] = sc.depth.view(np.uint8).reshape(-1)
                mc[:] = sc.color.reshape(-1)
                c.upload_mapped_frame()
            c.clear_occupied_bricks()
            c.process_textures()
        with pytest.raises(capi.RgbdrError):
            ranks[int(rng.integers(0, k))].integrate()              # the other ranks' sensors have not arrived
        views = [c.shard_view() for c in ranks]
        for c in ranks:
            c.sync()
        words = views[0].sensor_bytes // 4
        fr = [rdist.wrap_device_words(v.frames, words * n, dev) for v in views]
        cn = [rdist.wrap_device_words(v.counters, v.num_bricks, dev) for v in views]
        total = sum(t.clone() for t in cn)
        for r, v in enumerate(views):
            lo, hi = v.first * words, (v.first + v.count) * words
            for q in range(k):
                if q != r:
                    fr[q][lo:hi] = fr[r][lo:hi]
        for t in cn:
            t.copy_(total)
        torch.cuda.synchronize()
        for c in ranks:
            with pytest.raises(capi.RgbdrError):
                c.integrate()                       # looking at the buffers (shard_view) does not disarm the guard ...
            c.shard_gather_done()                   # ... the host's word that its own gather is enqueued does
        ref = orc.run_pipeline(sc, BMIN, BMAX, res, inv, limit=limit, brick_size=g.brick_size, bv=g.brick_voxels, res_bricks=tuple(g.res_bricks),
                               min_voxels=cfg.min_voxels_per_brick, use_bricks=bricks)
        for r, c in enumerate(ranks):
            c.update_occupied_bricks()
            c.integrate()
            assert np.array_equal(c.readback_brick_counters(), ref["counters"]), (desc, frame_no, r)
            assert np.array_equal(c.get_occupied()[0], ref["occupied"]), (desc, frame_no, r)
            got = c.readback_tsdf()
            assert same_bits(got, ref["tsdf"]), (desc, frame_no, r, bricks, pipelined, skip, count_diff(got, ref["tsdf"]))
            for i in range(r * (n // k), (r + 1) * (n // k)):     # the images of its own sensors
                assert same_bits(c.readback_image(IMG["quality"], i), ref["quality"][i]), (desc, frame_no, r, i)
    for c in ranks:
        c.close()
