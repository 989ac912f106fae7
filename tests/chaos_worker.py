"""Worker of tests/test_chaos_gpu.py: one chaos sequence in a process of its own, so that a call that hangs or crashes takes
this process down and not the test session; the call about to be made is written to the trace file first.
    python3 tests/chaos_worker.py <seed> <slab rank or -1> <slab count> <trace file>"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from conftest import load_oracle, load_package, same_bits  # noqa: E402

BMIN, BMAX = (-1.0, 0.0, -1.0), (1.0, 2.0, 1.0)


def run(pkg, orc, seed, slab, trace):
    capi, synth = pkg.capi, pkg.synth
    L = capi.lib()
    rng = np.random.default_rng(seed)
    n, W, H, G = 2, 64, 53, 32
    scene = synth.Scene(n, W, H, lut_res=(16, 13, 16))
    kw = dict(slab_rank=slab[0], slab_count=slab[1]) if slab else {}
    G = 48 if slab else 32
    ctx = capi.Context(capi.make_config(n, (W, H), voxel_size=2.0 / G, brick_size=8 * 2.0 / G, **kw), 0)
    h = ctx._h
    inv = scene.inverse((G, G, G))
    luts = {"xyz": [capi.make_lut(scene.xyz[i], (16, 13, 16)) for i in range(n)], "uv": [capi.make_lut(scene.uv[i], (16, 13, 16)) for i in range(n)],
            "inv": [capi.make_lut(inv[i], (G, G, G)) for i in range(n)]}
    depth = np.ascontiguousarray(scene.depth)
    color = np.ascontiguousarray(scene.color)
    big = np.zeros(4 * 1024 * 1024, np.float32)          # destination large enough for any readback of this context
    u64, u32, sz, f1, i1, i2, vp = C.c_uint64(), C.c_uint32(), C.c_size_t(), C.c_float(), C.c_int(), C.c_int(), C.c_void_p()
    F = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    U32 = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint32))
    sensor = lambda: int(rng.choice([-1, 0, 1, 2, 9, 1 << 30]))
    flag = lambda: int(rng.integers(-1, 3))

    def view():
        w, hh = [int(v) for v in rng.choice([0, 1, 7, 33, 64, 40000], 2)]
        v = capi.make_view((2.2, 1.6, 1.9), (0.0, 0.9, 0.0), (0.0, 1.0, 0.0), 50.0, max(w, 1), max(hh, 1), BMIN, BMAX,
                           shade_mode=int(rng.integers(0, 4)))
        v.width, v.height = (w, hh) if w * hh <= 64 * 64 else (int(rng.choice([0, 40000])), 5)     # only sizes `big` can take, or invalid ones
        v.skip_space = int(rng.integers(0, 2))
        if rng.integers(0, 6) == 0:
            v.shade_mode = 7
        if rng.integers(0, 5) == 0:                       # a poisoned uniform: refused, never marched
            member = str(rng.choice(["camera_pos", "modelview", "projection", "img_to_eye", "modelview_inv"]))
            getattr(v, member)[int(rng.integers(0, 3))] = float(rng.choice([float("nan"), float("inf"), -float("inf")]))
        return v

    ids = np.array([0, 1, 5, 2 ** 31], np.uint32)
    peer_buf = C.create_string_buffer(capi.HALO_PEER_BYTES)
    garbage = C.create_string_buffer(bytes(rng.integers(0, 256, capi.HALO_PEER_BYTES, dtype=np.uint8)), capi.HALO_PEER_BYTES)

    def peers_to_self():
        """export; a middle slab then becomes its own neighbour on BOTH sides (a face whose neighbour never writes would leave
        the side stream waiting for ever -- as a send without a receive does --, so a context stands in for one neighbour only
        if it also stands in for the other); an edge slab checks that the side it lacks is refused"""
        rc = L.rgbdr_halo_export(h, peer_buf)
        if rc == 0 and slab:
            if 0 < slab[0] < slab[1] - 1:
                assert L.rgbdr_halo_set_peer(h, 0, peer_buf) == 0 and L.rgbdr_halo_set_peer(h, 1, peer_buf) == 0
            else:
                assert L.rgbdr_halo_set_peer(h, 0 if slab[0] == 0 else 1, peer_buf) == capi.ERR_STATE
        return rc

    def no_peers():
        return min(L.rgbdr_halo_set_peer(h, 0, None), L.rgbdr_halo_set_peer(h, 1, None))

    calls = [
        lambda: L.rgbdr_draw(h, C.byref(view()), flag()), lambda: L.rgbdr_draw(h, None, 1),
        lambda: L.rgbdr_device_view_frame(h, flag(), C.byref(vp), C.byref(vp), C.byref(i1), C.byref(i2)),
        lambda: L.rgbdr_device_view_frame(h, 0, None, None, None, None),
        lambda: L.rgbdr_device_view_frame_async(h, flag(), C.byref(vp), C.byref(vp), C.byref(i1), C.byref(i2), C.byref(vp)),
        lambda: L.rgbdr_device_view_frame_async(h, flag(), None, None, None, None, None),
        lambda: L.rgbdr_readback_view_frame(h, flag(), F(big), F(big[1 << 20:])), lambda: L.rgbdr_readback_view_frame(h, 1, None, None),
        peers_to_self, lambda: L.rgbdr_halo_export(h, None),
        lambda: L.rgbdr_halo_set_peer(h, int(rng.integers(-1, 3)), garbage), no_peers,
        lambda: L.rgbdr_halo_pull_async(h),
        lambda: L.rgbdr_set_calibration(h, sensor(), C.byref(luts["xyz"][0]), C.byref(luts["uv"][0])),
        lambda: L.rgbdr_set_calibration(h, 0, None, C.byref(luts["uv"][0])),
        lambda: L.rgbdr_set_inverse_calibration(h, sensor(), C.byref(luts["inv"][0])),
        lambda: L.rgbdr_set_inverse_calibration(h, 0, None),
        lambda: L.rgbdr_load_calibration_files(h, sensor(), b"/nonexistent.cv_xyz", b"/nonexistent.cv_uv", None),
        lambda: L.rgbdr_load_calibration_files(h, 0, b"/nonexistent.cv_xyz", None, None),
        lambda: L.rgbdr_compute_inverse_calibration(h, sensor(), int(rng.integers(-2, 12))),
        lambda: L.rgbdr_generate_inverse_lut(h, sensor(), None, 2, None),
        lambda: L.rgbdr_upload_frame(h, depth.ctypes.data, color.ctypes.data),
        lambda: L.rgbdr_upload_frame(h, None, color.ctypes.data),
        lambda: L.rgbdr_upload_frame_device(h, None, None),
        lambda: L.rgbdr_upload_mapped_frame(h),
        lambda: L.rgbdr_map_frame_buffer(h, C.byref(vp), C.byref(vp), C.byref(sz), C.byref(sz)),
        lambda: L.rgbdr_map_frame_buffer(h, None, None, None, None),
        lambda: L.rgbdr_clear_occupied_bricks(h),
        lambda: L.rgbdr_process_textures(h),
        lambda: L.rgbdr_update_occupied_bricks(h),
        lambda: L.rgbdr_set_occupied_bricks(h, U32(ids), int(rng.integers(0, 5))),
        lambda: L.rgbdr_set_occupied_bricks(h, None, 3),
        lambda: L.rgbdr_integrate(h),
        lambda: L.rgbdr_step(h, depth.ctypes.data, color.ctypes.data),
        lambda: L.rgbdr_step(h, None, None),
        lambda: L.rgbdr_sync(h),
        lambda: L.rgbdr_set_tsdf_limit(h, float(rng.choice([0.01, 0.03, 0.0, -1.0, float("nan"), float("inf")]))),
        lambda: L.rgbdr_set_brick_size(h, float(rng.choice([8 * 2.0 / G, 5 * 2.0 / G, 0.0, float("nan"), float("inf"), 1e30]))),
        lambda: L.rgbdr_set_voxel_size(h, float(rng.choice([0.0, -1.0, float("nan"), float("inf"), 1e-9]))),      # (invalid only: keeps the grid)
        lambda: L.rgbdr_set_use_bricks(h, flag()), lambda: L.rgbdr_set_pipelined(h, flag()), lambda: L.rgbdr_set_elide_stores(h, flag()),
        lambda: L.rgbdr_set_skip_background(h, flag()), lambda: L.rgbdr_filter_textures(h, flag()),
        lambda: L.rgbdr_set_sweep_launches(h, int(rng.choice([1, 2, 7, 64, 0, -3, 65, 2 ** 31 - 1]))),
        lambda: L.rgbdr_use_processed_depths(h, flag()), lambda: L.rgbdr_refine_boundary(h, flag()),
        lambda: L.rgbdr_set_min_voxels_per_brick(h, int(rng.choice([0, 1, 10, 2 ** 32 - 1]))),
        lambda: L.rgbdr_skipped_pairs(h, C.byref(u64), C.byref(u64)), lambda: L.rgbdr_skipped_pairs(h, None, None),
        lambda: L.rgbdr_readback_skip_tables(h, int(rng.integers(-1, 4)), big.ctypes.data, int(rng.choice([0, 16, big.nbytes]))),
        lambda: L.rgbdr_get_geometry(h, None),
        lambda: L.rgbdr_get_camera_position(h, sensor(), F(big)),
        lambda: L.rgbdr_readback_tsdf(h, F(big)), lambda: L.rgbdr_readback_tsdf(h, None),
        lambda: L.rgbdr_readback_image(h, int(rng.integers(-1, 11)), sensor(), F(big)),
        lambda: L.rgbdr_readback_image(h, 2, 0, None),
        lambda: L.rgbdr_readback_inverse_calibration(h, sensor(), int(rng.integers(-2, 6)), int(rng.integers(-2, G + 3)), F(big)),
        lambda: L.rgbdr_readback_color(h, sensor(), big.ctypes.data_as(C.POINTER(C.c_uint8))),
        lambda: L.rgbdr_readback_brick_counters(h, U32(big.view(np.uint32))), lambda: L.rgbdr_readback_brick_counters(h, None),
        lambda: L.rgbdr_get_occupied(h, U32(big.view(np.uint32)), int(rng.choice([0, 1, 100000])), C.byref(sz), C.byref(f1)),
        lambda: L.rgbdr_get_occupied(h, None, 0, None, None),
        lambda: L.rgbdr_device_tsdf(h, None), lambda: L.rgbdr_device_frame(h, sensor(), C.byref(vp)),
        lambda: L.rgbdr_device_image(h, int(rng.integers(-1, 11)), sensor(), C.byref(capi.ImageDeviceView())),
        lambda: L.rgbdr_device_calibration(h, sensor(), C.byref(capi.CalibrationDeviceView())),
        lambda: L.rgbdr_device_calibration(h, 0, None),
        lambda: L.rgbdr_raymarch(h, C.byref(view()), F(big), F(big[1 << 20:]), F(big[2 << 20:])),
        lambda: L.rgbdr_raymarch(h, None, F(big), F(big), F(big)),
        lambda: L.rgbdr_raymarch_find(h, C.byref(view()), C.byref(vp)),
        lambda: L.rgbdr_raymarch_shade(h, C.byref(view()), F(big), F(big[1 << 20:]), F(big[2 << 20:])),
        lambda: L.rgbdr_draw_depth_limits(h, C.byref(view()), F(big)),
        lambda: L.rgbdr_fill_colors(h, F(big), F(big[1 << 20:])), lambda: L.rgbdr_fill_colors(h, None, None),
        lambda: L.rgbdr_upload_view_frame(h, int(rng.choice([0, 8, 40000])), 8, F(big), F(big)),
        lambda: L.rgbdr_upload_view_frame(h, 8, 8, None, F(big)),
        lambda: L.rgbdr_halo_staging(h, int(rng.integers(-1, 3)), C.byref(vp), C.byref(vp), C.byref(sz)),
        lambda: L.rgbdr_set_halo_staging(h, int(rng.integers(-2, 3))),
        lambda: L.rgbdr_readback_tile_layers(h, int(rng.integers(-1, 9)), int(rng.integers(-1, 3)), F(big)),
        lambda: L.rgbdr_halo_exchange(h, None, -1, -1, int(rng.integers(-1, 2)), None),
        lambda: L.rgbdr_halo_begin_step(h), lambda: L.rgbdr_halo_exchange_async(h, None, 0, 0), lambda: L.rgbdr_halo_wait(h),
        lambda: L.rgbdr_set_sensor_shard(h, int(rng.integers(-1, 3)), int(rng.integers(-1, 4))),
        lambda: L.rgbdr_shard_view(h, C.byref(capi.ShardDeviceView())), lambda: L.rgbdr_shard_view(h, None),
        lambda: L.rgbdr_shard_allgather(h, None), lambda: L.rgbdr_shard_gather_done(h), lambda: L.rgbdr_import_frame(h, None, None, None), lambda: L.rgbdr_import_frame_from(h, None), lambda: L.rgbdr_import_frame_from(h, h),
        lambda: L.rgbdr_shard_allgather_async(h, None),
        lambda: L.rgbdr_settle(h, 0.01, C.byref(f1)), lambda: L.rgbdr_settle(h, float("nan"), None),
        lambda: L.rgbdr_get_arena_probe(h, F(big), C.byref(i1), C.byref(i2)), lambda: L.rgbdr_get_arena_probe(h, None, None, None),
        lambda: L.rgbdr_get_arena_chunks(h, C.byref(i1), C.byref(f1)), lambda: L.rgbdr_get_arena_chunks(h, None, None),
        lambda: L.rgbdr_set_stream(h, None),
        lambda: L.rgbdr_enable_timers(h, flag()), lambda: L.rgbdr_enable_timer_accumulation(h, flag()),
        lambda: L.rgbdr_set_timer_detail(h, int(rng.integers(-1, 4))),
        lambda: L.rgbdr_timer_ns(h, rng.choice([b"2integrate", b"morph", b"nonsense", b""]), C.byref(u64)),
        lambda: L.rgbdr_timer_ns(h, None, None),
        lambda: L.rgbdr_timer_stats(h, rng.choice([b"2integrate", b"nonsense"]), C.byref(u64), C.byref(u32)),
    ]
    seen = set()
    for step_no in range(160):
        k = int(rng.integers(0, len(calls)))
        trace.write("%d %d\n" % (step_no, k))            # the parent reads the last line when this process hangs or dies
        trace.flush()
        rc = calls[k]()
        assert isinstance(rc, int) and -7 <= rc <= 0, (seed, step_no, k, rc)
        seen.add(rc)
        assert L.rgbdr_last_error(h) is not None
    assert 0 in seen and len(seen) >= 3                   # successes and several kinds of refusal
    # whatever is left: put the context into a defined state through the public calls and run a frame
    assert L.rgbdr_set_sensor_shard(h, 0, 0) == 0
    for i in range(n):
        ctx.set_calibration(i, scene.xyz[i], (16, 13, 16), scene.uv[i], (16, 13, 16), (0.5, 4.5))
        ctx.set_inverse_calibration(i, inv[i], (G, G, G))
    ctx.set_tsdf_limit(0.01)
    ctx.set_brick_size(8 * 2.0 / G)
    ctx.set_min_voxels_per_brick(10)
    for setter in (ctx.filter_textures, ctx.use_processed_depths, ctx.refine_boundary, ctx.set_use_bricks):
        setter(True)
    for setter in (ctx.set_pipelined, ctx.set_elide_stores, ctx.set_skip_background):
        setter(False)
    ctx.set_halo_staging(-1) if slab else None
    ctx.step(scene.depth, scene.color)
    g = ctx.geo
    ref = orc.run_pipeline(scene, BMIN, BMAX, tuple(g.res_volume), inv, limit=0.01, brick_size=g.brick_size, bv=g.brick_voxels,
                           res_bricks=tuple(g.res_bricks), min_voxels=10)
    assert same_bits(ctx.readback_tsdf(), ref["tsdf"][g.slab_voxel_z0:g.slab_voxel_z1]), seed
    assert np.array_equal(ctx.readback_brick_counters(), ref["counters"])
    ctx.close()


if __name__ == "__main__":
    load_package()
    from rgbd_recon_amd import capi, synth

    class P:
        pass
    pkg = P()
    pkg.capi, pkg.synth = capi, synth
    seed, r, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    with open(sys.argv[4], "w") as trace:
        run(pkg, load_oracle(), seed, (r, k) if r >= 0 else None, trace)
    print("chaos ok")
