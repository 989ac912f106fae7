"""The TSDF ray-marcher (glsl/tsdf_raymarch.fs + shading.glsl) on the
device against its oracle restatement: bit-exact colour, gl_FragDepth and sample
counts for every shade mode, from outside and inside the volume."""
import ctypes as C

import numpy as np
import pytest

from conftest import count_diff, same_bits

pytestmark = pytest.mark.gpu
BMIN, BMAX = (-1.0, 0.0, -1.0), (1.0, 2.0, 1.0)


def setup(pkg, orc, flags=15, inv_res=None, G=64, **cfg):
    capi, synth = pkg.capi, pkg.synth
    scene = synth.Scene(2, 128, 106, lut_res=(32, 27, 32))
    ctx = capi.Context(capi.make_config(2, (128, 106), voxel_size=2.0 / G, brick_size=8 * 2.0 / G, flags=flags, **cfg), 0)
    inv_res = inv_res or (G, G, G)
    inv = scene.inverse(inv_res)
    for i in range(2):
        ctx.set_calibration(i, scene.xyz[i], scene.lut_res, scene.uv[i], scene.lut_res, (0.5, 4.5))
        ctx.set_inverse_calibration(i, inv[i], inv_res)
    ctx.set_use_bricks(False)
    ctx.step(scene.depth, scene.color)
    return scene, ctx, inv


def oracle_images(orc, ctx, scene, inv_for_oracle, view):
    tsdf = ctx.readback_tsdf()
    db = [ctx.readback_image(4, i) for i in range(2)]
    q = [ctx.readback_image(7, i) for i in range(2)]
    return orc.raymarch(bytes(view), tsdf, inv_for_oracle, scene.uv, [scene.color[i] for i in range(2)], db, q)


@pytest.mark.parametrize("shade_mode", [0, 1, 2, 3])
@pytest.mark.parametrize("eye", [(2.2, 1.6, 1.9), (0.85, 1.7, 0.8)])      # outside / inside the box
def test_raymarch_matches_oracle(pkg, orc, shade_mode, eye):
    scene, ctx, inv = setup(pkg, orc)
    view = pkg.capi.make_view(eye, (0.0, 0.9, 0.0), (0.0, 1.0, 0.0), 50.0, 96, 72, BMIN, BMAX, shade_mode=shade_mode)
    color, depth, ns = ctx.raymarch(view)
    rc, rd, rn = oracle_images(orc, ctx, scene, inv, view)
    assert same_bits(ns, rn), count_diff(ns, rn)
    assert same_bits(depth, rd), count_diff(depth, rd)
    assert same_bits(color, rc), count_diff(color, rc)
    hit = depth < 1.0
    assert 0.02 < hit.mean() < 0.98                    # some rays hit the surface, some are discarded
    assert np.all(color[~hit] == np.float32([0, 1, 0, 0]))
    assert ns.max() > 0.1
    ctx.close()


@pytest.mark.parametrize("whole_wave", ["1", "2"])
def test_rays_marched_by_the_whole_wavefront(pkg, orc, monkeypatch, whole_wave):
    """the march hands a wavefront's last long rays to all of its 64 lanes, 64 samples a round (march_whole_wave): with every ray
    that is still marching after a round taken that way (RGBDR_WHOLE_WAVE_MARCH=1), and with none (2), the frames equal the
    oracle's -- through the whole cube and from the brick peels, in a viewport that leaves lanes of its wavefronts without a pixel"""
    monkeypatch.setenv("RGBDR_WHOLE_WAVE_MARCH", whole_wave)
    scene, ctx, inv = setup(pkg, orc)
    view = pkg.capi.make_view((2.2, 1.6, 1.9), (0.0, 0.9, 0.0), (0.0, 1.0, 0.0), 50.0, 130, 75, BMIN, BMAX)
    color, depth, ns = ctx.raymarch(view)
    rc, rd, rn = oracle_images(orc, ctx, scene, inv, view)
    assert same_bits(ns, rn) and same_bits(depth, rd) and same_bits(color, rc)
    assert (depth < 1.0).mean() > 0.02 and ns.max() / 0.0027 > 64          # rays of more than one round of the wavefront
    ctx.set_use_bricks(True)
    ctx.step(scene.depth, scene.color)
    g = ctx.geo
    ids, _ = ctx.get_occupied()
    mask = np.zeros(g.num_bricks, np.uint8)
    mask[ids] = 1
    peels = orc.depth_peels(bytes(view), BMIN, g.brick_size, tuple(g.res_bricks), ctx.readback_brick_counters(), mask)
    view.skip_space = 1
    color, depth, ns = ctx.raymarch(view)
    rc, rd, rn = oracle_images_skip(orc, ctx, scene, inv, view, peels)
    assert same_bits(ns, rn) and same_bits(depth, rd) and same_bits(color, rc)
    ctx.close()


@pytest.mark.parametrize("wh", [(1, 1), (7, 3), (17, 9), (63, 65), (129, 1)])
def test_viewports_off_the_block_size(pkg, orc, wh):
    """viewports of one pixel, narrower than the 8 x 8 square a wavefront covers, one row high, one past a block edge: the
    ray-march (with and without the brick depth peels) and the hole filling with its degenerate LOD pyramids equal the oracle"""
    scene, ctx, inv = setup(pkg, orc)
    view = pkg.capi.make_view((2.2, 1.6, 1.9), (0.0, 0.9, 0.0), (0.0, 1.0, 0.0), 50.0, wh[0], wh[1], BMIN, BMAX)
    color, depth, ns = ctx.raymarch(view)
    rc, rd, rn = oracle_images(orc, ctx, scene, inv, view)
    assert same_bits(ns, rn) and same_bits(depth, rd) and same_bits(color, rc)
    fc, fd = ctx.fill_colors(wh[0], wh[1])
    oc, od = orc.fill_colors(rc, rd)
    assert same_bits(fc, oc) and same_bits(fd, od)
    ctx.set_use_bricks(True)                                 # the occupied bricks of the frame, for the peels
    ctx.step(scene.depth, scene.color)
    g = ctx.geo
    peels = ctx.draw_depth_limits(view)
    ids, _ = ctx.get_occupied()
    mask = np.zeros(g.num_bricks, np.uint8)
    mask[ids] = 1
    ref = orc.depth_peels(bytes(view), BMIN, g.brick_size, tuple(g.res_bricks), ctx.readback_brick_counters(), mask)
    assert same_bits(peels, ref)
    view.skip_space = 1
    color, depth, ns = ctx.raymarch(view)
    rc, rd, rn = oracle_images_skip(orc, ctx, scene, inv, view, ref)
    assert same_bits(ns, rn) and same_bits(depth, rd) and same_bits(color, rc)
    ctx.close()


def test_depth_limits_of_another_viewport_leave_the_last_frame_alone(pkg, orc):
    """the reference draws its depth limits into their own FBO (m_view_depth), not into the window: a stand-alone
    rgbdr_draw_depth_limits -- of a smaller viewport, whose peels once landed inside the stored frame, or of a larger one,
    which once re-allocated it away -- must not change what rgbdr_fill_colors fills afterwards (found by
    test_random_view_sequences)"""
    scene, ctx, inv = setup(pkg, orc)
    ctx.set_use_bricks(True)
    ctx.step(scene.depth, scene.color)
    view = pkg.capi.make_view((2.2, 1.6, 1.9), (0.0, 0.9, 0.0), (0.0, 1.0, 0.0), 50.0, 48, 36, BMIN, BMAX)
    color, depth, _ = ctx.raymarch(view)
    want = orc.fill_colors(color, depth)
    for wh in ((33, 17), (64, 64), (200, 150), (48, 36)):
        other = pkg.capi.make_view((0.85, 1.7, 0.8), (0.0, 0.9, 0.0), (0.0, 1.0, 0.0), 50.0, wh[0], wh[1], BMIN, BMAX)
        ctx.draw_depth_limits(other)
        fc, fd = ctx.fill_colors(48, 36)
        assert same_bits(fc, want[0]) and same_bits(fd, want[1]), wh
    ctx.close()


@pytest.mark.parametrize("seed", list(range(1, 5 + int(__import__("os").environ.get("RGBDR_EXTRA_SEEDS", "0")) // 8)))
def test_random_view_sequences(pkg, orc, seed):
    """state machine of the consumer side: random interleavings of new frames (two scenes, both sweeps, another truncation
    limit), views of changing size / eye / shade mode, with and without space skipping, depth-limit draws, hole filling and
    externally uploaded view frames -- after every call the result equals the oracle's for the state in force (the view
    buffers are re-used and re-sized between calls, the peels and the brick table belong to the current frame)"""
    rng = np.random.default_rng(seed)
    capi, synth = pkg.capi, pkg.synth
    scene, ctx, inv = setup(pkg, orc, G=48)
    # 48 is no power of two: (x + 0.5) / 48 * 48 - 0.5 is not exactly x for every x, so the library keeps the LUT resampled at the
    # voxel centres (what tsdf_integration.vs looks up, ulps from the file's texels) and the ray-marcher samples THAT
    # (test_raymarch_with_resampled_lut; RGBDR_FLAG_NO_RESAMPLE keeps the file layout): the oracle gets the resident LUT
    inv = [ctx.readback_inverse_calibration(i, 0, 48) for i in range(2)]
    other = synth.Scene(2, 128, 106, lut_res=(32, 27, 32), seed=4321, sphere_r=0.7)
    scenes = [scene, other]
    cur, limit, bricks = 0, np.float32(0.01), False
    eyes = [(2.2, 1.6, 1.9), (0.85, 1.7, 0.8), (-2.0, 1.2, 2.1), (0.05, 1.95, 0.02)]
    sizes = [(48, 36), (33, 17), (64, 64), (9, 70), (96, 40)]
    last = None                                            # (colour, depth) of the last ray-marched / uploaded frame
    for step_no in range(14):
        op = int(rng.integers(0, 6))
        if op == 0 or step_no == 0:                        # a new frame
            cur = int(rng.integers(0, 2))
            bricks = bool(rng.integers(0, 2))
            limit = np.float32(rng.choice([0.01, 0.03]))
            ctx.set_use_bricks(bricks)
            ctx.set_tsdf_limit(float(limit))
            ctx.step(scenes[cur].depth, scenes[cur].color)
            last = None
            continue
        sc = scenes[cur]
        w, h = sizes[int(rng.integers(0, len(sizes)))]
        view = capi.make_view(eyes[int(rng.integers(0, len(eyes)))], (0.0, 0.9, 0.0), (0.0, 1.0, 0.0), 50.0, w, h, BMIN, BMAX,
                              shade_mode=int(rng.integers(0, 4)))
        g = ctx.geo
        if op in (1, 2):                                   # ray-march, with the peels when the frame has an occupied list
            peels = None
            if op == 2 and bricks:
                ids, _ = ctx.get_occupied()
                mask = np.zeros(g.num_bricks, np.uint8)
                mask[ids] = 1
                peels = orc.depth_peels(bytes(view), BMIN, g.brick_size, tuple(g.res_bricks), ctx.readback_brick_counters(), mask)
                view.skip_space = 1
            color, depth, ns = ctx.raymarch(view)
            tsdf = ctx.readback_tsdf()
            db = [ctx.readback_image(4, i) for i in range(2)]
            q = [ctx.readback_image(7, i) for i in range(2)]
            # (the calibration is `scene`'s whichever scene the frame's images came from)
            rc, rd, rn = orc.raymarch(bytes(view), tsdf, inv, scene.uv, [sc.color[i] for i in range(2)], db, q, limit=float(limit), peels=peels)
            assert same_bits(ns, rn) and same_bits(depth, rd) and same_bits(color, rc), (seed, step_no, op, (w, h))
            last = (rc, rd)
        elif op == 3 and bricks:                           # the depth limits alone
            ids, _ = ctx.get_occupied()
            mask = np.zeros(g.num_bricks, np.uint8)
            mask[ids] = 1
            ref = orc.depth_peels(bytes(view), BMIN, g.brick_size, tuple(g.res_bricks), ctx.readback_brick_counters(), mask)
            assert same_bits(ctx.draw_depth_limits(view), ref), (seed, step_no, op, (w, h))
        elif op == 4 and last is not None:                 # hole filling of the last frame (whatever was drawn or peeled since)
            fc, fd = ctx.fill_colors(last[1].shape[1], last[1].shape[0])
            oc, od = orc.fill_colors(*last)
            assert same_bits(fc, oc) and same_bits(fd, od), (seed, step_no, op, last[1].shape)
        elif op == 5:                                      # a frame composited elsewhere becomes the last frame
            col = rng.random((h, w, 4), dtype=np.float32)
            dep = np.where(rng.random((h, w)) < 0.4, np.float32(1.0), rng.random((h, w), dtype=np.float32)).astype(np.float32)
            col[dep >= 1] = np.float32([0, 1, 0, 0])
            ctx.upload_view_frame(col, dep)
            last = (col, dep)
    ctx.close()


def test_raymarch_with_file_layout_lut(pkg, orc):
    """RGBDR_FLAG_NO_RESAMPLE: colours are looked up through the file-resolution LUT"""
    scene, ctx, inv = setup(pkg, orc, flags=15 | pkg.capi.FLAG_NO_RESAMPLE, inv_res=(45, 50, 45))
    view = pkg.capi.make_view((2.2, 1.6, 1.9), (0.0, 0.9, 0.0), (0.0, 1.0, 0.0), 50.0, 80, 60, BMIN, BMAX)
    color, depth, ns = ctx.raymarch(view)
    rc, rd, rn = oracle_images(orc, ctx, scene, inv, view)
    assert same_bits(color, rc) and same_bits(depth, rd) and same_bits(ns, rn)
    ctx.close()


def test_raymarch_with_resampled_lut(pkg, orc):
    """default handling of a non-1:1 LUT: the ray-marcher samples the resident
    (resampled, grid-layout) LUT -- equal to the oracle fed with that LUT"""
    scene, ctx, inv = setup(pkg, orc, inv_res=(45, 50, 45))
    resident = [ctx.readback_inverse_calibration(i, 0, 64) for i in range(2)]
    view = pkg.capi.make_view((2.2, 1.6, 1.9), (0.0, 0.9, 0.0), (0.0, 1.0, 0.0), 50.0, 80, 60, BMIN, BMAX)
    color, depth, ns = ctx.raymarch(view)
    rc, rd, rn = oracle_images(orc, ctx, scene, resident, view)
    assert same_bits(color, rc) and same_bits(depth, rd) and same_bits(ns, rn)
    ctx.close()


@pytest.mark.parametrize("mode", [1, 5])
def test_raymarch_samples_dxt_colour_frames_as_uploaded(pkg, orc, mode):
    """compress_rgb 1 / 5 (the reference's default is DXT1): the shading's colour lookup decodes its four texels from the
    blocks on the spot -- the frame the GL driver would have decoded into the texture (squish's arithmetic, orc.decode_dxt) --
    without the whole-frame decode; once a consumer has asked for the RGB8 frame the lookup reads that: the same frame, bit for
    bit, either way"""
    capi, synth = pkg.capi, pkg.synth
    scene, ctx, inv = setup(pkg, orc, compress_rgb=mode)
    W, H = 128, 106
    blocks = np.stack([synth.encode_dxt(scene.color[i], mode) for i in range(2)])
    rng = np.random.default_rng(11)
    blocks[1, : blocks.shape[1] // 4] = rng.integers(0, 256, blocks.shape[1] // 4, dtype=np.uint8)   # arbitrary blocks too
    decoded = [orc.decode_dxt(blocks[i], W, H, mode) for i in range(2)]
    view = capi.make_view((2.2, 1.6, 1.9), (0.0, 0.9, 0.0), (0.0, 1.0, 0.0), 50.0, 96, 72, BMIN, BMAX)
    ctx.step(scene.depth, blocks)
    color, depth, ns = ctx.raymarch(view)                       # from the blocks
    tsdf = ctx.readback_tsdf()
    db = [ctx.readback_image(4, i) for i in range(2)]
    q = [ctx.readback_image(7, i) for i in range(2)]
    rc, rd, rn = orc.raymarch(bytes(view), tsdf, inv, scene.uv, decoded, db, q)
    assert same_bits(ns, rn) and same_bits(depth, rd) and same_bits(color, rc), count_diff(color, rc)
    assert (rd < 1).mean() > 0.05 and len(np.unique(color[rd < 1][:, :3])) > 50
    assert np.array_equal(ctx.readback_color(1), decoded[1])    # now the RGB8 frame exists
    c2, d2, n2 = ctx.raymarch(view)                             # ... and is what the lookup reads
    assert same_bits(c2, rc) and same_bits(d2, rd) and same_bits(n2, rn)
    ctx.close()


def test_raymarch_errors(pkg, orc):
    capi = pkg.capi
    scene, ctx, inv = setup(pkg, orc, G=32)
    view = capi.make_view((2.2, 1.6, 1.9), (0.0, 0.9, 0.0), (0.0, 1.0, 0.0), 50.0, 32, 24, BMIN, BMAX)
    view.shade_mode = 7
    with pytest.raises(capi.RgbdrError):
        ctx.raymarch(view)
    view.shade_mode = 0
    for w, h in ((0, 24), (32, -1), (32768 + 1, 24), (2 ** 31 - 1, 2 ** 31 - 1)):     # a viewport nothing can be (sizes would wrap)
        view.width, view.height = w, h
        ptr = C.c_void_p()
        assert capi.lib().rgbdr_raymarch_find(ctx._h, C.byref(view), C.byref(ptr)) == capi.ERR_INVALID_ARGUMENT
        assert capi.lib().rgbdr_draw_depth_limits(ctx._h, C.byref(view), None) == capi.ERR_INVALID_ARGUMENT
    view.width, view.height = 32, 24
    # a NaN / infinite uniform would turn the sample count of tsdf_raymarch.fs:86 into the largest unsigned: refused, not marched
    for poison in (float("nan"), float("inf"), float("-inf")):
        for member, k in (("camera_pos", 0), ("modelview", 5), ("img_to_eye", 15), ("projection", 10), ("vol_to_world_inv", 0)):
            bad = capi.View.from_buffer_copy(bytes(view))
            getattr(bad, member)[k] = poison
            ptr = C.c_void_p()
            assert capi.lib().rgbdr_raymarch_find(ctx._h, C.byref(bad), C.byref(ptr)) == capi.ERR_INVALID_ARGUMENT, (member, poison)
            assert capi.lib().rgbdr_draw_depth_limits(ctx._h, C.byref(bad), None) == capi.ERR_INVALID_ARGUMENT
    ctx.close()
    ctx2 = capi.Context(capi.make_config(1, (64, 53), voxel_size=2.0 / 32, brick_size=0.5), 0)
    with pytest.raises(capi.RgbdrError) as e:
        ctx2.raymarch(view)
    assert e.value.status == capi.ERR_STATE
    ctx2.close()


# sizes: LOD 1 on its own launch + a tail (96 x 72, 130 x 75, 641 x 359), the reference's window (1280 x 720: three launches, then
# the tail from LOD 5), the whole pyramid in the tail workgroup (33 x 17, 9 x 6), no pyramid at all (3 x 1), one LOD (2 x 2, 7 x 3)
@pytest.mark.parametrize("wh", [(96, 72), (130, 75), (641, 359), (1280, 720), (33, 17), (9, 6), (7, 3), (2, 2), (3, 1), (70, 2)])
def test_fill_colors_matches_oracle(pkg, orc, wh):
    scene, ctx, inv = setup(pkg, orc)
    view = pkg.capi.make_view((2.2, 1.6, 1.9), (0.0, 0.9, 0.0), (0.0, 1.0, 0.0), 50.0, wh[0], wh[1], BMIN, BMAX)
    color, depth, _ = ctx.raymarch(view)
    fc, fd = ctx.fill_colors(wh[0], wh[1])
    rc, rd = orc.fill_colors(color, depth)
    assert same_bits(fd, rd), count_diff(fd, rd)
    assert same_bits(fc, rc), count_diff(fc, rc)
    holes = color[..., 3] <= 0
    if min(wh) >= 17:
        assert holes.mean() > 0.05 and (fc[..., 3] > 0)[holes].mean() > 0.5     # holes exist and get filled
    if wh == (96, 72):        # (for sizes where res * (px / res) rounds below px the shader reads a neighbour)
        assert same_bits(fc[~holes], color[~holes])
    ctx.close()


@pytest.mark.parametrize("wh", [(1280, 720), (1279, 721), (257, 130), (64, 64), (31, 200), (200, 31), (5, 4), (4096, 3)])
@pytest.mark.parametrize("hole_share", [0.02, 0.6, 0.97])
def test_fill_colors_of_random_frames(pkg, orc, wh, hole_share):
    """frames composited elsewhere (rgbdr_upload_view_frame) with random holes, alpha of every sign, depth on both sides of 1: the
    pyramid without its two atlases (taps folded through framebuffer_transfer.fs's index map, LOD 0 read from the frame, the tail
    LODs in one workgroup) equals the oracle's ping-pong of whole atlases bit for bit -- colour and depth"""
    capi = pkg.capi
    rng = np.random.default_rng(wh[0] * 7919 + wh[1] + int(hole_share * 100))
    w, h = wh
    ctx = capi.Context(capi.make_config(1, (64, 53), voxel_size=2.0 / 32, brick_size=0.5), 0)
    col = rng.random((h, w, 4), dtype=np.float32)
    col[..., 3] = np.where(rng.random((h, w)) < 0.1, np.float32(-1.0), col[..., 3])       # tsdf_inpaint.fs writes alpha -1 itself
    dep = rng.random((h, w), dtype=np.float32)
    # holes in blobs (so that whole 4 x 4 neighbourhoods of the coarse LODs are empty) and as salt
    coarse = rng.random(((h + 15) // 16, (w + 15) // 16)) < hole_share
    hole = np.kron(coarse, np.ones((16, 16), bool))[:h, :w] | (rng.random((h, w)) < 0.05)
    col[hole] = np.float32([0, 1, 0, 0])
    dep[hole] = np.where(rng.random(int(hole.sum())) < 0.5, np.float32(1.0), np.float32(0.25))   # depth < 1 under alpha 0: the -1 branch
    ctx.upload_view_frame(col, dep)
    fc, fd = ctx.fill_colors(w, h)
    rc, rd = orc.fill_colors(col, dep)
    assert same_bits(fd, rd), count_diff(fd, rd)
    assert same_bits(fc, rc), count_diff(fc, rc)
    fc2, fd2 = ctx.fill_colors(w, h)                  # again on the same buffers: nothing of the first run is read
    assert same_bits(fc2, rc) and same_bits(fd2, rd)
    ctx.close()


def test_an_upload_ahead_of_the_draw_on_a_pipelined_context(pkg, orc):
    """the next frame may be uploaded before the current one is drawn: a pipelined context keeps the colour frame in two halves, so
    the view pass still shades with the colours of the frame it shows (a sequential context has one half: there the draw would
    see the new colours, as before); with a zero-copy view of the colour image handed out the uploads stay in the half the
    view points to and the pass waits for them instead"""
    capi, synth = pkg.capi, pkg.synth
    a = synth.Scene(2, 128, 106, lut_res=(32, 27, 32), color_wh=(128, 106))
    b = synth.Scene(2, 128, 106, lut_res=(32, 27, 32), seed=4321, sphere_r=0.7, color_wh=(128, 106))
    inv = a.inverse((64, 64, 64))
    view = capi.make_view((2.2, 1.6, 1.9), (0.0, 0.9, 0.0), (0.0, 1.0, 0.0), 50.0, 160, 90, BMIN, BMAX)

    def context(pipelined):
        ctx = capi.Context(capi.make_config(2, (128, 106), voxel_size=2.0 / 64, brick_size=8 * 2.0 / 64), 0)
        for i in range(2):
            ctx.set_calibration(i, a.xyz[i], a.lut_res, a.uv[i], a.lut_res, (0.5, 4.5))
            ctx.set_inverse_calibration(i, inv[i], (64, 64, 64))
        ctx.set_pipelined(pipelined)
        return ctx

    def frame(ctx, sc, ahead=None):
        ctx.update(sc.depth, sc.color)
        ctx.clear_occupied_bricks(); ctx.process_textures(); ctx.update_occupied_bricks(); ctx.integrate()
        if ahead is not None:
            ctx.update(ahead.depth, ahead.color)
        ctx.draw(view, False)
        return ctx.readback_view_frame(False)

    seq, pipe = context(False), context(True)
    want_a = frame(seq, a)
    for k in range(3):                                       # (both halves take their turn)
        got = frame(pipe, a, ahead=b)
        assert same_bits(got[0], want_a[0]) and same_bits(got[1], want_a[1]), k
    want_b = frame(seq, b)
    got = frame(pipe, b)
    assert same_bits(got[0], want_b[0]) and same_bits(got[1], want_b[1])
    assert not same_bits(want_a[0], want_b[0])
    v0 = pipe.device_image(capi.IMG_COLOR, 0)                # from here on the colour frame stays where this view points
    for k in range(3):
        sc, want = ((a, want_a), (b, want_b))[k % 2]
        got = frame(pipe, sc)
        assert same_bits(got[0], want[0]) and same_bits(got[1], want[1]), k
        assert pipe.device_image(capi.IMG_COLOR, 0).ptr == v0.ptr
    seq.close()
    pipe.close()


@pytest.mark.parametrize("compress_rgb", [0, 1])
def test_a_host_far_ahead_of_a_pipelined_context(pkg, orc, compress_rgb):
    """sixteen frames of two alternating scenes enqueued without the host ever waiting (frames uploaded from device memory, every
    ray-marched frame copied aside by a device-to-device copy on the context's stream): the chain of frame k + 2 refills the
    buffers frame k lives in, and must not do so before the view pass of frame k has read them -- every frame equals the
    sequential context's"""
    import ctypes as C
    import torch
    capi, synth = pkg.capi, pkg.synth
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    scenes = [synth.Scene(2, 128, 106, lut_res=(32, 27, 32), color_wh=(128, 106)),
              synth.Scene(2, 128, 106, lut_res=(32, 27, 32), seed=4321, sphere_r=0.7, color_wh=(128, 106))]
    colours = [np.stack([synth.encode_dxt(sc.color[i], 1) for i in range(2)]) if compress_rgb else sc.color for sc in scenes]
    dev = [(torch.from_numpy(sc.depth).cuda(), torch.from_numpy(np.ascontiguousarray(c)).cuda()) for sc, c in zip(scenes, colours)]
    inv = scenes[0].inverse((64, 64, 64))
    view = capi.make_view((2.2, 1.6, 1.9), (0.0, 0.9, 0.0), (0.0, 1.0, 0.0), 50.0, 640, 360, BMIN, BMAX)
    view.skip_space = 1
    K, npix = 16, 640 * 360
    frames = []
    for pipelined in (False, True):
        ctx = capi.Context(capi.make_config(2, (128, 106), voxel_size=2.0 / 64, brick_size=8 * 2.0 / 64, compress_rgb=compress_rgb), 0)
        for i in range(2):
            ctx.set_calibration(i, scenes[0].xyz[i], scenes[0].lut_res, scenes[0].uv[i], scenes[0].lut_res, (0.5, 4.5))
            ctx.set_inverse_calibration(i, inv[i], (64, 64, 64))
        ctx.set_use_bricks(True)
        ctx.set_pipelined(pipelined)
        stream = capi.lib().rgbdr_stream(ctx._h)
        got = [torch.empty(npix * 5, dtype=torch.float32, device="cuda") for _ in range(K)]
        torch.cuda.synchronize()
        for k in range(K):
            d, c = dev[k % 2]
            ctx.update_device(d.data_ptr(), c.data_ptr())
            ctx.clear_occupied_bricks(); ctx.process_textures(); ctx.update_occupied_bricks(); ctx.integrate()
            ctx.draw(view, True)
            cp, dp, w, h = ctx.device_view_frame(False)
            assert hip.hipMemcpyAsync(C.c_void_p(got[k].data_ptr()), C.c_void_p(cp), npix * 16, 3, C.c_void_p(stream)) == 0
            assert hip.hipMemcpyAsync(C.c_void_p(got[k].data_ptr() + npix * 16), C.c_void_p(dp), npix * 4, 3, C.c_void_p(stream)) == 0
        ctx.sync()
        frames.append([g.cpu().numpy() for g in got])
        ctx.close()
    for k in range(K):
        assert same_bits(frames[0][k], frames[1][k]), (k, count_diff(frames[0][k], frames[1][k]))
    assert not same_bits(frames[0][0], frames[0][1])


def test_a_presenter_on_its_own_queue(pkg, orc):
    """rgbdr_device_view_frame_async: a host that presents from its own queue takes the pointers of frame k and an event, enqueues
    frame k + 1 at once, and lets its queue wait for the event before it copies the frame out -- nothing on the context's
    streams waits for it.  Ten frames of two alternating scenes on a pipelined context (hole filling on its own stream, filled
    image and view buffers in two halves): every frame the presenter copied equals the one a sequential context shows."""
    import ctypes as C
    import torch
    capi, synth = pkg.capi, pkg.synth
    hip = C.CDLL("libamdhip64.so")
    hip.hipStreamWaitEvent.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
    hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    scenes = [synth.Scene(2, 128, 106, lut_res=(32, 27, 32), color_wh=(128, 106)),
              synth.Scene(2, 128, 106, lut_res=(32, 27, 32), seed=4321, sphere_r=0.7, color_wh=(128, 106))]
    inv = scenes[0].inverse((64, 64, 64))
    view = capi.make_view((2.2, 1.6, 1.9), (0.0, 0.9, 0.0), (0.0, 1.0, 0.0), 50.0, 320, 180, BMIN, BMAX)
    view.skip_space = 1
    ctxs = []
    for pipelined in (False, True):
        ctx = capi.Context(capi.make_config(2, (128, 106), voxel_size=2.0 / 64, brick_size=8 * 2.0 / 64), 0)
        for i in range(2):
            ctx.set_calibration(i, scenes[0].xyz[i], scenes[0].lut_res, scenes[0].uv[i], scenes[0].lut_res, (0.5, 4.5))
            ctx.set_inverse_calibration(i, inv[i], (64, 64, 64))
        ctx.set_use_bricks(True)
        ctx.set_pipelined(pipelined)
        ctxs.append(ctx)

    def enqueue(ctx, k):
        sc = scenes[k % 2]
        ctx.update(sc.depth, sc.color)
        ctx.clear_occupied_bricks(); ctx.process_textures(); ctx.update_occupied_bricks(); ctx.integrate()
        ctx.draw(view, True)

    K, npix = 10, 320 * 180
    want = []
    for k in range(K):
        enqueue(ctxs[0], k)
        want.append(ctxs[0].readback_view_frame(True))
    present = torch.cuda.Stream()
    got = [(torch.empty(npix * 4, dtype=torch.float32, device="cuda"), torch.empty(npix, dtype=torch.float32, device="cuda")) for _ in range(K)]
    copied = [None] * K
    for k in range(K):
        if k >= 2:
            copied[k - 2].synchronize()                      # the presenter is done with frame k - 2: its half is free again
        enqueue(ctxs[1], k)
        c, d, w, h, ev = ctxs[1].device_view_frame_async(True)
        assert (w, h) == (320, 180) and ev
        assert hip.hipStreamWaitEvent(C.c_void_p(present.cuda_stream), C.c_void_p(ev), 0) == 0
        assert hip.hipMemcpyAsync(C.c_void_p(got[k][0].data_ptr()), C.c_void_p(c), npix * 16, 3, C.c_void_p(present.cuda_stream)) == 0
        assert hip.hipMemcpyAsync(C.c_void_p(got[k][1].data_ptr()), C.c_void_p(d), npix * 4, 3, C.c_void_p(present.cuda_stream)) == 0
        copied[k] = torch.cuda.Event()
        copied[k].record(present)
    present.synchronize()
    for k in range(K):
        assert same_bits(got[k][0].cpu().numpy().reshape(180, 320, 4), want[k][0]), k
        assert same_bits(got[k][1].cpu().numpy().reshape(180, 320), want[k][1]), k
    # ... and on the sequential context the event is behind what its stream holds
    c, d, w, h, ev = ctxs[0].device_view_frame_async(True)
    assert ev and c == ctxs[0].device_view_frame(True)[0]
    for ctx in ctxs:
        ctx.close()


@pytest.mark.parametrize("skip", [False, True])
def test_draw_is_drawF_without_a_host_round_trip(pkg, orc, skip):
    """rgbdr_draw = ReconIntegration::drawF (recon_integration.cpp:151-178): depth limits when skipping, the ray-march, fillColors
    -- enqueued behind the frame's passes with no host synchronisation, the frame left on the device.  Frame after frame (two
    scenes alternating, so that a draw that overtook its integrate would show the other scene) the device frame equals the
    oracle's ray-march and its hole filling bit for bit."""
    capi, synth = pkg.capi, pkg.synth
    scene, ctx, inv = setup(pkg, orc)
    other = synth.Scene(2, 128, 106, lut_res=(32, 27, 32), seed=4321, sphere_r=0.7)
    ctx.set_use_bricks(skip)
    view = capi.make_view((2.2, 1.6, 1.9), (0.0, 0.9, 0.0), (0.0, 1.0, 0.0), 50.0, 160, 90, BMIN, BMAX)
    view.skip_space = 1 if skip else 0
    with pytest.raises(capi.RgbdrError) as e:                # nothing drawn yet
        ctx.readback_view_frame(False)
    assert e.value.status == capi.ERR_STATE
    g = ctx.geo
    for k in range(4):
        sc = (scene, other)[k % 2]
        ctx.update(sc.depth, sc.color)                       # the frame's calls and the draw: nothing waits in between
        ctx.clear_occupied_bricks()
        ctx.process_textures()
        ctx.update_occupied_bricks()
        ctx.integrate()
        ctx.draw(view, fill_holes=(k != 1))
        if k == 1:
            with pytest.raises(capi.RgbdrError) as e:        # this frame was not filled: the filled image belongs to the last one
                ctx.readback_view_frame(True)
            assert e.value.status == capi.ERR_STATE
        color, depth = ctx.readback_view_frame(False)
        tsdf = ctx.readback_tsdf()
        db = [ctx.readback_image(4, i) for i in range(2)]
        q = [ctx.readback_image(7, i) for i in range(2)]
        peels = None
        if skip:
            ids, _ = ctx.get_occupied()
            mask = np.zeros(g.num_bricks, np.uint8)
            mask[ids] = 1
            peels = orc.depth_peels(bytes(view), BMIN, g.brick_size, tuple(g.res_bricks), ctx.readback_brick_counters(), mask)
        rc, rd, _ = orc.raymarch(bytes(view), tsdf, inv, scene.uv, [sc.color[i] for i in range(2)], db, q, peels=peels)
        assert same_bits(depth, rd) and same_bits(color, rc), (k, count_diff(color, rc))
        if k != 1:
            fc, fd = ctx.readback_view_frame(True)
            oc, od = orc.fill_colors(rc, rd)
            assert same_bits(fc, oc) and same_bits(fd, od), (k, count_diff(fc, oc))
            cptr, dptr, w, h = ctx.device_view_frame(True)
            assert (w, h) == (160, 90) and cptr and dptr and cptr != ctx.device_view_frame(False)[0]
    assert (rd < 1).mean() > 0.05
    ctx.close()

@pytest.mark.parametrize("compress_rgb", [0, 1])
def test_draw_on_a_pipelined_context(pkg, orc, compress_rgb):
    """a context with RGBDR_FLAG_PIPELINE runs the pre_* chain of frame k + 1 on its second stream while rgbdr_draw of frame k
    is still on the first: the view pass is ordered by events (no host synchronisation), the next upload waits for its last
    read of the colour frame.  Bursts of 2 .. 6 frames of two alternating scenes, nothing read back inside a burst: the last
    displayed frame of every burst equals, bit for bit, the one a sequential context displays for the same frames."""
    capi, synth = pkg.capi, pkg.synth
    scenes = [synth.Scene(2, 128, 106, lut_res=(32, 27, 32), color_wh=(128, 106)),
              synth.Scene(2, 128, 106, lut_res=(32, 27, 32), seed=4321, sphere_r=0.7, color_wh=(128, 106))]
    frames = [(sc.depth, np.stack([synth.encode_dxt(sc.color[i], 1) for i in range(2)]) if compress_rgb else sc.color) for sc in scenes]
    inv = scenes[0].inverse((64, 64, 64))
    ctxs = []
    for pipelined in (False, True):
        ctx = capi.Context(capi.make_config(2, (128, 106), voxel_size=2.0 / 64, brick_size=8 * 2.0 / 64, compress_rgb=compress_rgb), 0)
        for i in range(2):
            ctx.set_calibration(i, scenes[0].xyz[i], scenes[0].lut_res, scenes[0].uv[i], scenes[0].lut_res, (0.5, 4.5))
            ctx.set_inverse_calibration(i, inv[i], (64, 64, 64))
        ctx.set_use_bricks(True)
        ctx.set_pipelined(pipelined)
        ctxs.append(ctx)
    view = capi.make_view((2.2, 1.6, 1.9), (0.0, 0.9, 0.0), (0.0, 1.0, 0.0), 50.0, 320, 180, BMIN, BMAX)
    view.skip_space = 1
    k = 0
    for burst in (2, 3, 4, 5, 6):
        shown = []
        for ctx in ctxs:
            kk = k
            for _ in range(burst):
                depth, color = frames[kk % 2]
                kk += 1
                ctx.update(depth, color)
                ctx.clear_occupied_bricks()
                ctx.process_textures()
                ctx.update_occupied_bricks()
                ctx.integrate()
                ctx.draw(view, True)
            shown.append(ctx.readback_view_frame(True) + ctx.readback_view_frame(False))
        k += burst
        for a, b in zip(*shown):
            assert same_bits(a, b), (burst, count_diff(a, b))
        assert (shown[0][3] < 1).mean() > 0.05
    bad = capi.View.from_buffer_copy(bytes(view))          # a view the library refuses leaves the last frame where it was
    bad.shade_mode = 7
    for ctx in ctxs:
        with pytest.raises(capi.RgbdrError):
            ctx.draw(bad, True)
    again = [ctx.readback_view_frame(True) + ctx.readback_view_frame(False) for ctx in ctxs]
    for a, b, c in zip(again[0], again[1], shown[0]):
        assert same_bits(a, b) and same_bits(a, c)
    # the hole filling of a pipelined context runs on a stream of its own (the next frame's sweep and march under it): a
    # fill asked for on the context's stream (rgbdr_fill_colors) comes after it, viewports of another size re-make the buffers
    # under it, and switching the pipeline off drains it
    for wh in ((320, 180), (96, 54), (640, 360), (320, 180)):
        v2 = capi.make_view((2.2, 1.6, 1.9), (0.0, 0.9, 0.0), (0.0, 1.0, 0.0), 50.0, wh[0], wh[1], BMIN, BMAX)
        v2.skip_space = 1
        shown = []
        for ctx in ctxs:
            for kk in (k, k + 1, k + 2):
                depth, color = frames[kk % 2]
                ctx.update(depth, color)
                ctx.clear_occupied_bricks(); ctx.process_textures(); ctx.update_occupied_bricks(); ctx.integrate()
                ctx.draw(v2, True)
            shown.append(ctx.fill_colors(wh[0], wh[1]) + ctx.readback_view_frame(True))
        k += 3
        for a, b in zip(*shown):
            assert same_bits(a, b), (wh, count_diff(a, b))
        assert same_bits(shown[1][0], shown[1][2]) and same_bits(shown[1][1], shown[1][3])      # the two fills agree
    ctxs[1].set_pipelined(False)
    for ctx in ctxs:
        depth, color = frames[k % 2]
        ctx.update(depth, color)
        ctx.clear_occupied_bricks(); ctx.process_textures(); ctx.update_occupied_bricks(); ctx.integrate()
        ctx.draw(view, True)
    a, b = ctxs[0].readback_view_frame(True), ctxs[1].readback_view_frame(True)
    assert same_bits(a[0], b[0]) and same_bits(a[1], b[1])
    for ctx in ctxs:
        ctx.close()


def test_fill_colors_needs_a_frame(pkg):
    capi = pkg.capi
    ctx = capi.Context(capi.make_config(1, (64, 53), voxel_size=2.0 / 32, brick_size=0.5), 0)
    with pytest.raises(capi.RgbdrError) as e:
        ctx.fill_colors(8, 8)
    assert e.value.status == capi.ERR_STATE
    ctx.close()


@pytest.mark.parametrize("eye", [(2.2, 1.6, 1.9), (0.85, 1.7, 0.8), (0.05, 1.95, 0.02)])
def test_depth_peels_and_space_skipping(pkg, orc, eye):
    """drawDepthLimits + the skipSpace start positions of the ray-marcher (f-4)"""
    scene, ctx, inv = setup(pkg, orc)
    ctx.set_use_bricks(True)
    ctx.step(scene.depth, scene.color)
    g = ctx.geo
    view = pkg.capi.make_view(eye, (0.0, 0.9, 0.0), (0.0, 1.0, 0.0), 50.0, 96, 72, BMIN, BMAX)
    peels = ctx.draw_depth_limits(view)
    counters = ctx.readback_brick_counters()
    ids, _ = ctx.get_occupied()
    mask = np.zeros(g.num_bricks, np.uint8)
    mask[ids] = 1
    ref = orc.depth_peels(bytes(view), BMIN, g.brick_size, tuple(g.res_bricks), counters, mask)
    assert same_bits(peels, ref), count_diff(peels, ref)
    touched = peels[..., 0] < 1.0
    assert 0.02 < touched.mean() < 0.98
    assert np.all(peels[~touched] == np.float32([1, 0, 1, 0]))
    assert np.all(-peels[touched][:, 1] >= peels[touched][:, 0])            # farthest >= nearest
    # ray-march with space skipping == oracle with the same peels; surfaces agree with the full march
    view.skip_space = 1
    color, depth, ns = ctx.raymarch(view)
    rc, rd, rn = oracle_images_skip(orc, ctx, scene, inv, view, ref)
    assert same_bits(ns, rn) and same_bits(depth, rd) and same_bits(color, rc)
    view.skip_space = 0
    _, depth_full, ns_full = ctx.raymarch(view)
    hit_skip, hit_full = depth < 1, depth_full < 1
    assert (hit_skip & ~hit_full).mean() < 0.01
    both = hit_skip & hit_full
    assert both.mean() > 0.02 and np.percentile(np.abs(depth[both] - depth_full[both]), 99) < 5e-3
    assert ns.sum() < 0.8 * ns_full.sum()                                     # fewer samples marched
    ctx.close()


def oracle_images_skip(orc, ctx, scene, inv, view, peels):
    tsdf = ctx.readback_tsdf()
    db = [ctx.readback_image(4, i) for i in range(2)]
    q = [ctx.readback_image(7, i) for i in range(2)]
    return orc.raymarch(bytes(view), tsdf, inv, scene.uv, [scene.color[i] for i in range(2)], db, q, peels=peels)


@pytest.mark.parametrize("count,G,tsdf_limit", [(2, 64, 0.03), (3, 64, 0.03), (4, 128, 0.03), (2, 96, 0.1), (3, 50, 0.03), (2, 100, 0.05)])
@pytest.mark.parametrize("skip", [0, 1])
def test_slab_raymarch_equals_whole_volume(pkg, orc, count, G, tsdf_limit, skip):
    """Z slabs: find -> element-wise MIN of the first-hit indices -> shade -> select.
    The slab contexts live on one GPU here; the halo layers and the MIN that RCCL
    carries between ranks (rgbd_recon_amd.dist) are moved with torch copies."""
    import torch

    from rgbd_recon_amd import dist as rdist

    dev = torch.device("cuda:0")
    scene, whole, inv = setup(pkg, orc, tsdf_limit=tsdf_limit, G=G)
    ctxs = [setup(pkg, orc, tsdf_limit=tsdf_limit, G=G, slab_rank=r, slab_count=count)[1] for r in range(count)]
    if skip:
        for c in [whole] + ctxs:
            c.set_use_bricks(True)
            c.step(scene.depth, scene.color)
    halo = ctxs[0].geo.halo_tile_layers
    assert halo == -(-int(np.ceil(np.float32(tsdf_limit) * np.float32(G)) + 2) // 8)
    views = []
    for c in ctxs:
        c.sync()
        views.append(rdist.halo_views(c.device_tsdf(), dev))       # send_lo, send_hi, recv_lo, recv_hi
    for r in range(count - 1):
        views[r + 1][2].copy_(views[r][1])                           # r's top layers -> (r+1)'s lower halo
        views[r][3].copy_(views[r + 1][0])                           # (r+1)'s bottom layers -> r's upper halo
    torch.cuda.synchronize()
    for shade_mode, eye in [(0, (2.2, 1.6, 1.9)), (1, (0.85, 1.7, 0.8)), (2, (0.1, 1.2, 2.4))]:
        view = pkg.capi.make_view(eye, (0.0, 0.9, 0.0), (0.0, 1.0, 0.0), 50.0, 96, 72, BMIN, BMAX, shade_mode=shade_mode)
        view.skip_space = skip
        ref_c, ref_d, ref_n = whole.raymarch(view)
        npix = view.width * view.height
        ks = [rdist.wrap_device_int32(c.raymarch_find(view), npix, dev) for c in ctxs]
        kmin = torch.stack(ks).min(dim=0).values
        owners_with_hits = sum(int((k != rdist.NO_HIT).any()) for k in ks)
        for k in ks:
            k.copy_(kmin)
        torch.cuda.synchronize()
        color = np.tile(np.float32([0, 1, 0, 0]), (view.height, view.width, 1))
        depth = np.ones((view.height, view.width), np.float32)
        shaded = np.zeros((view.height, view.width), np.int32)
        for c, k in zip(ctxs, ks):
            cc, dd, nn = c.raymarch_shade(view)
            assert same_bits(nn, ref_n)                              # every slab knows the sample count
            mine = (k.cpu().numpy() != rdist.NO_HIT).reshape(view.height, view.width)
            color[mine], depth[mine] = cc[mine], dd[mine]
            shaded += mine
            assert np.all(dd[~mine] == 1.0) and np.all(cc[~mine] == np.float32([0, 1, 0, 0]))
        assert shaded.max() == 1                                     # exactly one owner per hit pixel
        assert same_bits(depth, ref_d), count_diff(depth, ref_d)
        assert same_bits(color, ref_c), count_diff(color, ref_c)
        assert (ref_d < 1).mean() > 0.02
        if shade_mode == 0:
            assert owners_with_hits >= 2                             # the surface really spans slabs
    with pytest.raises(pkg.capi.RgbdrError):
        ctxs[0].raymarch(view)                                       # a slab cannot march alone
    for c in [whole] + ctxs:
        c.close()
