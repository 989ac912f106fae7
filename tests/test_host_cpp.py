"""The C++ host mirror (rgbd-recon_amd/host/rgbdr_host.hpp: CalibVolumes /
NetKinectArray / ReconIntegration over the C ABI) driven by the example frame
loop, fed with calibration volumes and a recorded .stream frame in the reference's
on-disk formats."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import count_diff, same_bits

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "rgbd-recon_amd", "host", "frame_loop")


def test_frame_loop_is_built_and_linked_against_the_c_abi():
    assert os.path.exists(EXE), "run __graft_entry__.build()"
    out = subprocess.run(["ldd", EXE], capture_output=True, text=True).stdout
    assert "librgbdr_hip.so" in out and "not found" not in out
    assert subprocess.run([EXE], capture_output=True).returncode == 2      # usage
    inv = os.path.join(ROOT, "rgbd-recon_amd", "host", "calib_inverter")
    assert os.path.exists(inv), "run __graft_entry__.build()"
    assert subprocess.run([inv], capture_output=True).returncode == 2      # usage


@pytest.mark.gpu
def test_frame_loop_matches_oracle(pkg, orc, tmp_path):
    synth = pkg.synth
    n, W, H, G = 2, 64, 53, 32
    scene = synth.Scene(n, W, H, lut_res=(16, 13, 16), seed=77)
    inv = scene.inverse((G, G, G))
    d = str(tmp_path)
    os.makedirs(os.path.join(d, "recordings"))
    for i in range(n):
        assert orc.lut_write(os.path.join(d, "s%d.cv_xyz" % i), scene.xyz[i], 3) == 0
        assert orc.lut_write(os.path.join(d, "s%d.cv_uv" % i), scene.uv[i], 2) == 0
        assert orc.lut_write(os.path.join(d, "s%d.cv_xyz_inv" % i), inv[i], 4) == 0
        # .stream: frames of [colorsize][depthsize] (NetKinectArray.cpp:745-763); two frames, the first is read
        with open(os.path.join(d, "recordings", "s%d.stream" % i), "wb") as f:
            for k in range(2):
                f.write(scene.color[i].tobytes())
                f.write((scene.depth[i] if k == 0 else np.zeros_like(scene.depth[i])).tobytes())
    out = os.path.join(d, "out.tsdf")
    view = pkg.capi.make_view((2.2, 1.6, 1.9), (0.0, 0.9, 0.0), (0.0, 1.0, 0.0), 50.0, 48, 36, synth.BBOX_MIN, synth.BBOX_MAX)
    with open(os.path.join(d, "view.bin"), "wb") as f:
        f.write(bytes(view))
    r = subprocess.run([EXE, d, str(n), str(W), str(H), str(G), out, os.path.join(d, "view.bin")], capture_output=True,
                       text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout.startswith("res 32 32 32 bricks 64")
    r_first_stdout = r.stdout
    # Reconstruction::setColorMaskMode through the base pointer (anaglyph path: red, then green + blue of the same pixels)
    import re
    m = re.search(r"color masks: (\d+) pixels hit, (\d+) channel values wrong", r.stdout)
    assert m and int(m.group(1)) > 30 and int(m.group(2)) == 0, r.stdout
    got = np.fromfile(out, dtype=np.float32).reshape(G, G, G)
    # ReconIntegration::drawF through the C++ mirror == the same calls through the ctypes harness
    frame = np.fromfile(out + ".frame", dtype=np.float32)
    ctx = pkg.capi.Context(pkg.capi.make_config(n, (W, H), voxel_size=2.0 / G, brick_size=8 * 2.0 / G), 0)
    for i in range(n):
        ctx.set_calibration(i, scene.xyz[i], scene.lut_res, scene.uv[i], scene.lut_res, (0.5, 4.5))
        ctx.set_inverse_calibration(i, inv[i], (G, G, G))
    ctx.step(scene.depth, scene.color)
    view.skip_space = 1                                        # m_skip_space defaults to true
    ctx.raymarch(view)
    fc, fd = ctx.fill_colors(view.width, view.height)          # m_fill_holes defaults to true
    ctx.close()
    # frame() is the reference's window: the colorfill fragments of rays that hit nothing (depth exactly 1) fail its
    # GL_LESS depth test and the window keeps its clear colour, zeros (seen in the run of the shaders on Mesa)
    fc = np.where((fd < 1)[..., None], fc, np.float32(0.0))
    assert same_bits(frame[: fc.size].reshape(fc.shape), fc) and same_bits(frame[fc.size:].reshape(fd.shape), fd)
    assert 0.02 < (fd < 1).mean() < 0.98
    # ... and == the oracle end to end: g_recons.at(mode)->drawF() through the Reconstruction base pointer in
    # frame_loop.cpp is depth peels -> getStartPos -> ray-march -> inpaint pyramid -> colorfill of the reference
    gg = pkg.capi.compute_geometry(pkg.capi.make_config(n, (W, H), voxel_size=2.0 / G, brick_size=8 * 2.0 / G))
    ref_all = orc.run_pipeline(scene, synth.BBOX_MIN, synth.BBOX_MAX, (G, G, G), inv, brick_size=gg.brick_size,
                               bv=gg.brick_voxels, res_bricks=tuple(gg.res_bricks))
    peels = orc.depth_peels(bytes(view), synth.BBOX_MIN, gg.brick_size, tuple(gg.res_bricks), ref_all["counters"], ref_all["mask"])
    oc, od, _ = orc.raymarch(bytes(view), ref_all["tsdf"], inv, scene.uv, [scene.color[i] for i in range(n)],
                             ref_all["depth_b"], ref_all["quality"], peels=peels)
    foc, fod = orc.fill_colors(oc, od)
    foc = np.where((fod < 1)[..., None], foc, np.float32(0.0))
    assert same_bits(frame[: fc.size].reshape(fc.shape), foc), count_diff(frame[: fc.size].reshape(fc.shape), foc)
    assert same_bits(frame[fc.size:].reshape(fd.shape), fod)
    g = pkg.capi.compute_geometry(pkg.capi.make_config(n, (W, H), voxel_size=2.0 / G, brick_size=8 * 2.0 / G))
    ref = orc.run_pipeline(scene, synth.BBOX_MIN, synth.BBOX_MAX, (G, G, G), inv, brick_size=g.brick_size,
                           bv=g.brick_voxels, res_bricks=tuple(g.res_bricks))
    assert same_bits(got, ref["tsdf"]), "%d voxels differ" % count_diff(got, ref["tsdf"])
    # the second drawing mode of frame_loop.cpp (ReconFrameConsumer: the reference's constructor triple, held in the same
    # vector, drawn through the Reconstruction base pointer) consumed colour / depth_b / quality / normals of every sensor
    # zero-copy -- the four images ReconTrigrid samples (recon_trigrid.cpp:30-33): their digests are those of the oracle's images
    m = re.search(r"frame images((?: [0-9a-f]{16})+)", r_first_stdout)
    assert m, r_first_stdout
    digests = [int(h, 16) for h in m.group(1).split()]

    def fnv(a):
        # FNV-1a 64 over the bytes, vectorised: h = (h ^ b) * p is affine in h, so fold blocks with prefix products
        h, p, mask = 1469598103934665603, 1099511628211, (1 << 64) - 1
        for b in np.ascontiguousarray(a).view(np.uint8).reshape(-1).tolist():
            h = ((h ^ b) * p) & mask
        return h
    want = []
    for i in range(n):
        want += [fnv(scene.color[i]), fnv(ref["depth_b"][i]), fnv(ref["quality"][i]), fnv(ref["normal"][i])]
    assert digests == want, "the C++ consumer read other images than the oracle's"
    # ... and the calibration volumes the other modes sample on CalibVolumes' units (recon_calibs.cpp:35-36), where they live
    # on the device: the records of the files, cv_xyz without the padding lane of the device layout; getVolumeRes() is the
    # inverse volume's resolution as in the reference (CalibVolumes.cpp:90-92), getDepthLimits the file header's
    m = re.search(r"calibration volumes((?: [0-9a-f]{16})+) inv_res (\d+) (\d+) (\d+) limits (\S+) (\S+)", r_first_stdout)
    assert m, r_first_stdout
    want = []
    for i in range(n):
        want += [fnv(np.ascontiguousarray(scene.xyz[i], dtype=np.float32)), fnv(np.ascontiguousarray(scene.uv[i], dtype=np.float32))]
    assert [int(h, 16) for h in m.group(1).split()] == want, "the C++ consumer read other calibration volumes than the files'"
    assert [int(m.group(k)) for k in (2, 3, 4)] == [G, G, G] and (float(m.group(5)), float(m.group(6))) == (0.5, 4.5)
    # the same frame as one server message (K1 colour, K1 depth, K2 colour, ...; NetKinectArray.cpp:511-541)
    msg = os.path.join(d, "message.bin")
    with open(msg, "wb") as f:
        for i in range(n):
            f.write(scene.color[i].tobytes())
            f.write(scene.depth[i].tobytes())
    out2 = os.path.join(d, "out2.tsdf")
    r = subprocess.run([EXE, d, str(n), str(W), str(H), str(G), out2], capture_output=True, text=True,
                       env=dict(os.environ, RGBDR_MESSAGE_FILE=msg))
    assert r.returncode == 0, r.stderr
    assert same_bits(np.fromfile(out2, dtype=np.float32).reshape(G, G, G), got)
    stamp = np.frombuffer(scene.color[0].tobytes()[:8], dtype=np.float64)[0]      # the first 8 bytes double as the time
    assert ("frame time %.17g" % stamp) in r.stderr


@pytest.mark.gpu
def test_timer_database_keeps_the_reference_statistics_and_csv_files(pkg, orc, tmp_path):
    """host::TimerDatabase over the library's timers: duration(name) per frame like the application's GUI reads it
    (kinect_client.cpp:431-481), the fold of timer_database.cpp:26-41 in sample(), and the three CSV files the application
    writes when it quits (:835-851) -- names in map order, the configuration's name in front of the data line, values in ms"""
    import re
    synth = pkg.synth
    n, W, H, G = 2, 64, 53, 32
    scene = synth.Scene(n, W, H, lut_res=(16, 13, 16), seed=77)
    inv = scene.inverse((G, G, G))
    d = str(tmp_path)
    os.makedirs(os.path.join(d, "recordings"))
    for i in range(n):
        assert orc.lut_write(os.path.join(d, "s%d.cv_xyz" % i), scene.xyz[i], 3) == 0
        assert orc.lut_write(os.path.join(d, "s%d.cv_uv" % i), scene.uv[i], 2) == 0
        assert orc.lut_write(os.path.join(d, "s%d.cv_xyz_inv" % i), inv[i], 4) == 0
        with open(os.path.join(d, "recordings", "s%d.stream" % i), "wb") as f:
            f.write(scene.color[i].tobytes())
            f.write(scene.depth[i].tobytes())
    view = pkg.capi.make_view((2.2, 1.6, 1.9), (0.0, 0.9, 0.0), (0.0, 1.0, 0.0), 50.0, 48, 36, synth.BBOX_MIN, synth.BBOX_MAX)
    with open(os.path.join(d, "view.bin"), "wb") as f:
        f.write(bytes(view))
    csv = os.path.join(d, "scene,2026-10-4,3-5.csv")            # <conf>,<date>,<time>.csv like kinect_client.cpp:840-847
    r = subprocess.run([EXE, d, str(n), str(W), str(H), str(G), os.path.join(d, "out.tsdf"), os.path.join(d, "view.bin")],
                       capture_output=True, text=True, env=dict(os.environ, RGBDR_TIMER_CSV=csv))
    assert r.returncode == 0, r.stderr
    names = ["1preprocess", "2integrate", "3recon", "bilateral", "boundary", "brickdraw", "draw", "holefill", "morph", "normal", "quality"]
    m = re.search(r"^timers (.*)$", r.stdout, flags=re.M)
    assert m, r.stdout
    tok = m.group(1).split()
    last = {tok[4 * k]: (float(tok[4 * k + 1]), float(tok[4 * k + 2]), float(tok[4 * k + 3])) for k in range(len(names))}
    assert sorted(last) == names
    for name, (dur, mean, num) in last.items():
        assert dur > 0 and mean > 0 and num == 4, (name, dur, mean, num)             # every pass ran in each of the four frames
    assert abs(last["3recon"][0] - (last["brickdraw"][0] + last["draw"][0] + last["holefill"][0])) < 1.0
    assert last["1preprocess"][0] >= last["bilateral"][0]
    rows = {}
    for kind in ("mean", "min", "max"):
        text = open(os.path.join(d, kind + "_scene,2026-10-4,3-5.csv")).read().splitlines()
        assert text[0] == "timer," + ",".join('"%s"' % n for n in names), text[0]
        cells = text[1].split(",")
        assert cells[0] == "scene" and len(cells) == 1 + len(names) and len(text) == 2
        rows[kind] = [float(c) for c in cells[1:]]
    for k, name in enumerate(names):
        assert rows["min"][k] <= rows["mean"][k] * 1.000001, name
        assert abs(rows["mean"][k] - last[name][1] / 1e6) <= 1e-5 * rows["mean"][k] + 1e-9, name       # ms, six significant digits in the file (half a unit of the sixth)
        # the reference's `else if`: the first sample only lowers the minimum, so a timer whose first sample was its largest
        # keeps max 0 -- min <= max holds only when a later sample exceeded an earlier one
        assert rows["max"][k] == 0.0 or rows["max"][k] >= rows["min"][k], name


@pytest.mark.gpu
def test_frame_loop_from_ks_scene_with_reference_defaults(pkg, orc, tmp_path):
    """`.ks` + sensor `.yml` + LUT files + `.stream` recordings exactly as the
    reference lays them out, at its default operating point (voxel 0.01, brick 0.1,
    bbox up to y = 2.2, DXT1 colour, inverse LUT finer than the grid)"""
    synth = pkg.synth
    n, W, H = 2, 64, 52
    scene = synth.Scene(n, W, H, lut_res=(16, 13, 16), seed=55)
    bmin, bmax = (-1.0, 0.0, -1.0), (1.0, 2.2, 1.0)
    inv_res = (90, 99, 90)
    inv = [synth.inverse_lut(s, inv_res, bmin, bmax) for s in scene.sensors]
    d = str(tmp_path)
    os.makedirs(os.path.join(d, "calib"))
    os.makedirs(os.path.join(d, "recordings"))
    blocks = [synth.encode_dxt(scene.color[i], 1) for i in range(n)]
    with open(os.path.join(d, "scene.ks"), "w") as f:
        f.write("serverport 127.0.0.1:7000\n")
        for i in range(n):
            f.write("kinect calib/k%d.yml\n" % i)
        f.write("bbx -1.0 0.0 -1.0 1.0 2.2 1.0\n")
    for i in range(n):
        with open(os.path.join(d, "calib", "k%d.yml" % i), "w") as f:
            f.write("%%YAML:1.0\nrgb_size: [ %d, %d ]\ndepth_size: [ %d, %d ]\nnear_far: [ 0.5, 4.5 ]\n" % (W, H, W, H))
            if i == 0:
                f.write("compress_rgb: [ 1, 0 ]\ncompress_depth: [ 0, 0 ]\n")
        assert orc.lut_write(os.path.join(d, "calib", "k%d.cv_xyz" % i), scene.xyz[i], 3) == 0
        assert orc.lut_write(os.path.join(d, "calib", "k%d.cv_uv" % i), scene.uv[i], 2) == 0
        assert orc.lut_write(os.path.join(d, "k%d.cv_xyz_inv" % i), inv[i], 4) == 0
        with open(os.path.join(d, "recordings", "k%d.stream" % i), "wb") as f:
            f.write(blocks[i].tobytes())
            f.write(scene.depth[i].tobytes())
    out = os.path.join(d, "out.tsdf")
    r = subprocess.run([EXE, "--ks", os.path.join(d, "scene.ks"), "0.01", out], capture_output=True, text=True, cwd=d)
    assert r.returncode == 0, r.stderr
    assert r.stdout.startswith("sensors 2 depth 64x52 color 64x52 compress_rgb 1 compress_depth 0 res 200 221 200")
    got = np.fromfile(out, dtype=np.float32).reshape(200, 221, 200)
    cfg = pkg.capi.make_config(n, (W, H), bbox_min=bmin, bbox_max=bmax, voxel_size=0.01, brick_size=0.1)
    g = pkg.capi.compute_geometry(cfg)

    class Decoded:
        pass

    s2 = Decoded()
    s2.__dict__.update(scene.__dict__)
    s2.color = np.stack([orc.decode_dxt(blocks[i], W, H, 1) for i in range(n)])
    ref = orc.run_pipeline(s2, bmin, bmax, (200, 221, 200), inv, brick_size=g.brick_size, bv=g.brick_voxels,
                           res_bricks=tuple(g.res_bricks))
    assert same_bits(got, ref["tsdf"]), "%d voxels differ" % count_diff(got, ref["tsdf"])
    assert np.sum(np.abs(got) < 0.01) > 1000


@pytest.mark.gpu
def test_calib_inverter_tool(pkg, orc, tmp_path):
    """the reference's offline tool `calib_inverter <scene.ks> -s <voxel>` over the C ABI: same
    command line, file names and file format; the volumes equal rgbdr_generate_inverse_lut"""
    synth = pkg.synth
    n, W, H = 2, 64, 52
    scene = synth.Scene(n, W, H, lut_res=(16, 13, 16), seed=5, make_frames=False)
    d = str(tmp_path)
    os.makedirs(os.path.join(d, "calib"))
    with open(os.path.join(d, "scene.ks"), "w") as f:
        for i in range(n):
            f.write("kinect calib/k%d.yml\n" % i)
        f.write("bbx -1.0 0.0 -1.0 1.0 2.2 1.0\n")
    for i in range(n):
        with open(os.path.join(d, "calib", "k%d.yml" % i), "w") as f:
            f.write("%%YAML:1.0\nrgb_size: [ %d, %d ]\ndepth_size: [ %d, %d ]\nnear_far: [ 0.5, 4.5 ]\n" % (W, H, W, H))
        assert orc.lut_write(os.path.join(d, "calib", "k%d.cv_xyz" % i), scene.xyz[i], 3) == 0
        assert orc.lut_write(os.path.join(d, "calib", "k%d.cv_uv" % i), scene.uv[i], 2) == 0
    exe = os.path.join(ROOT, "rgbd-recon_amd", "host", "calib_inverter")
    r = subprocess.run([exe, os.path.join(d, "scene.ks"), "-s", "0.05"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    res = (40, 44, 40)                                         # ceil((2, 2.2, 2) / 0.05)
    assert "using resolution 40, 44, 40" in r.stdout
    ctx = pkg.capi.Context(pkg.capi.make_config(n, (W, H), bbox_min=(-1.0, 0.0, -1.0), bbox_max=(1.0, 2.2, 1.0),
                                                voxel_size=2.2 / 32, brick_size=2.2 / 4), 0)
    for i in range(n):
        ctx.set_calibration(i, scene.xyz[i], scene.lut_res, scene.uv[i], scene.lut_res, (0.5, 4.5))
        got, limits = orc.lut_read(os.path.join(d, "k%d.cv_xyz_inv" % i), 4)
        assert got.shape == (res[2], res[1], res[0], 4) and tuple(limits) == (0.5, 4.5)
        want = ctx.generate_inverse_lut(i, res)
        assert same_bits(got, want)
        assert 0.02 < (got[..., 3] == 1.0).mean() < 0.98       # inside / outside the frustum
    ctx.close()
    assert subprocess.run([exe, os.path.join(d, "scene.txt")], capture_output=True).returncode == 1   # "No .ks file specified"


SLAB_EXE = os.path.join(ROOT, "rgbd-recon_amd", "host", "slab_loop")


def test_slab_loop_is_built_against_rccl_and_the_c_abi():
    assert os.path.exists(SLAB_EXE), "run __graft_entry__.build()"
    out = subprocess.run(["ldd", SLAB_EXE], capture_output=True, text=True).stdout
    assert "librgbdr_hip.so" in out and "librccl" in out and "not found" not in out
    assert subprocess.run([SLAB_EXE], capture_output=True).returncode == 2      # usage
    # the library itself has no link-time dependency on RCCL (it is bound at run time)
    lib = subprocess.run(["ldd", os.path.join(ROOT, "rgbd-recon_amd", "librgbdr_hip.so")], capture_output=True, text=True).stdout
    assert "librccl" not in lib


@pytest.mark.gpu
@pytest.mark.parametrize("n,lag", [(2, False), (1, False), (2, True)])
def test_cpp_halo_exchanger_over_rccl_loopback(pkg, orc, tmp_path, n, lag):
    """n = 2: the pre_* chain goes through host::FrameGather on a communicator of its own; n = 1: a single sensor cannot be
    sharded, the loop runs the whole chain and creates no gather (it used to throw there); lag: host::LaggedChain --
    a chain-only backend one frame ahead, rgbdr_shard_allgather_async on its own stream, rgbdr_import_frame_from, flush at the
    end -- must leave the same faces.
    The C++ host's multi-GPU frame loop (host/slab_loop.cpp: host::HaloExchanger over
    rgbdr_halo_begin_step / _exchange_async / _wait, RCCL bound at run time by the library) as an
    inner Z slab whose two neighbours are the process itself: four different frames with no host
    synchronisation in between; afterwards the halo layers hold the boundary layers of the LAST frame."""
    capi, synth = pkg.capi, pkg.synth
    W, H, G, frames = 128, 106, 96, 4
    scenes = [synth.Scene(n, W, H, lut_res=(32, 27, 32), seed=1 + k, sphere_r=0.9 - 0.05 * k) for k in range(frames)]
    first = scenes[0]
    inv = first.inverse((G, G, G))
    d = str(tmp_path)
    os.makedirs(os.path.join(d, "recordings"))
    for i in range(n):
        assert orc.lut_write(os.path.join(d, "s%d.cv_xyz" % i), first.xyz[i], 3) == 0
        assert orc.lut_write(os.path.join(d, "s%d.cv_uv" % i), first.uv[i], 2) == 0
        assert orc.lut_write(os.path.join(d, "s%d.cv_xyz_inv" % i), inv[i], 4) == 0
        with open(os.path.join(d, "recordings", "s%d.stream" % i), "wb") as f:
            for k in range(frames):
                f.write(scenes[k].color[i].tobytes())
                f.write(scenes[k].depth[i].tobytes())
    out = os.path.join(d, "halo.bin")
    r = subprocess.run([SLAB_EXE, d, str(n), str(W), str(H), str(G), str(frames), out, "--loopback"] + (["--lag"] if lag else []), capture_output=True,
                       text=True, timeout=600, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    raw = np.fromfile(out, dtype=np.uint8)
    h, layer = np.frombuffer(raw[:8].tobytes(), dtype=np.int32)
    body = np.frombuffer(raw[8:8 + 16 * h * layer].tobytes(), dtype=np.float32).reshape(4, h * layer)
    ms = np.frombuffer(raw[8 + 16 * h * layer:].tobytes(), dtype=np.float64)[0]
    halo_lo, face_lo, face_hi, halo_hi = body
    # loopback: the lower face comes back through the "neighbour below" into the lower halo, the upper into the upper
    assert same_bits(halo_lo, face_lo) and same_bits(halo_hi, face_hi)
    assert ms > 0
    # the faces are those of the LAST frame: slab 1 of 4 of a context fed the frames one by one
    cfg = capi.make_config(n, (W, H), voxel_size=2.0 / G, brick_size=8 * 2.0 / G, slab_rank=1, slab_count=4)
    ctx = capi.Context(cfg, 0)
    assert ctx.geo.halo_tile_layers == h
    for i in range(n):
        ctx.set_calibration(i, first.xyz[i], first.lut_res, first.uv[i], first.lut_res, (0.5, 4.5))
        ctx.set_inverse_calibration(i, inv[i], (G, G, G))
    ctx.set_use_bricks(False)
    faces = []
    for k in range(frames):
        ctx.step(scenes[k].depth, scenes[k].color)
        v = ctx.device_tsdf()
        lo = np.empty(h * layer, np.float32)
        hi = np.empty(h * layer, np.float32)
        ctx._chk(capi.lib().rgbdr_readback_tile_layers(ctx._h, h, h, lo.ctypes.data_as(C.POINTER(C.c_float))))
        ctx._chk(capi.lib().rgbdr_readback_tile_layers(ctx._h, v.owned_layers, h, hi.ctypes.data_as(C.POINTER(C.c_float))))
        faces.append((lo, hi))
    ctx.close()
    assert same_bits(face_lo, faces[-1][0]) and same_bits(face_hi, faces[-1][1])
    assert not same_bits(faces[-1][0], faces[-2][0])             # the frames differ at the faces
    assert np.nanmax(np.abs(face_lo)) > 0
