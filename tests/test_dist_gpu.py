"""Several ranks, one Z slab each, end to end through rgbd_recon_amd.dist: integrate,
halo exchange, slab ray-march with first-hit MIN and frame compositing.  The box has
one GPU, so the ranks share cuda:0 and use gloo with host staging; on a multi-GPU node
the same functions run over RCCL (bench.py --gpus N)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import same_bits

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world,G,limit,backend", [(2, 64, 0.03, "gloo"), (3, 96, 0.05, "gloo"), (1, 64, 0.03, "nccl")])
def test_slab_ranks_render_the_whole_volume(world, G, limit, backend, tmp_path):
    """(the nccl case has one rank -- the box has one GPU -- but sends the library's device
    buffers, wrapped without a copy, through RCCL's all-reduce MIN / SUM)"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "slab_worker.py"), str(tmp_path), str(G),
           str(limit), backend]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    for n in range(2):
        ref = np.load(os.path.join(str(tmp_path), "whole_v%d.npz" % n))
        assert (ref["depth"] < 1).mean() > 0.02
        for rank in range(world):
            got = np.load(os.path.join(str(tmp_path), "slab_r%d_v%d.npz" % (rank, n)))
            for key in ("color", "depth", "ns"):
                assert same_bits(got[key], ref[key]), (n, rank, key)


@pytest.mark.parametrize("mode", ["exchanger", "exchanger_lib"])
@pytest.mark.parametrize("world", [2, 3])
def test_async_halo_exchanger(world, mode, tmp_path, pkg):
    """HaloExchanger: staged, stream-ordered exchange over three different frames with no
    host synchronisation in between; the halos end up holding the neighbours' boundary
    layers of the LAST frame, and the slabs the whole volume of that frame"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    G = 64
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "slab_worker.py"), mode, str(tmp_path),
           str(G)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    z = [np.load(os.path.join(str(tmp_path), "halo_r%d.npz" % k)) for k in range(world)]
    for k in range(world - 1):
        up = [same_bits(z[k]["recv_hi"], z[k + 1]["hist_lo"][f]) for f in range(3)]
        down = [same_bits(z[k + 1]["recv_lo"], z[k]["hist_hi"][f]) for f in range(3)]
        assert up[2] and down[2], (k, up, down)                  # the halos hold the last frame
        assert same_bits(z[k]["recv_hi"], z[k + 1]["send_lo"])
        assert same_bits(z[k + 1]["recv_lo"], z[k]["send_hi"])
        assert np.nanmax(np.abs(z[k]["send_hi"])) > 0
    # the last frame on one context
    capi, synth = pkg.capi, pkg.synth
    last = synth.Scene(2, 128, 106, lut_res=(32, 27, 32), seed=3, sphere_r=0.75)
    first = synth.Scene(2, 128, 106, lut_res=(32, 27, 32), seed=1, sphere_r=0.9)
    ctx = capi.Context(capi.make_config(2, (128, 106), voxel_size=2.0 / G, brick_size=8 * 2.0 / G), 0)
    inv = first.inverse((G, G, G))
    for i in range(2):
        ctx.set_calibration(i, first.xyz[i], first.lut_res, first.uv[i], first.lut_res, (0.5, 4.5))
        ctx.set_inverse_calibration(i, inv[i], (G, G, G))
    ctx.step(last.depth, last.color)
    assert same_bits(np.concatenate([zz["tsdf"] for zz in z], axis=0), ctx.readback_tsdf())
    ctx.close()


def run_worker(world, *args):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "slab_worker.py")] + [str(a) for a in args]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("world", [2, 3])
def test_halo_by_copy_engine_between_processes(world, tmp_path, pkg):
    """rgbdr_halo_export / rgbdr_halo_set_peer / rgbdr_halo_pull_async: the slabs are PROCESSES sharing the GPU, every rank
    pulls its neighbours' staged faces through HIP IPC mappings behind interprocess events -- three different frames, both
    sweeps, no host synchronisation in between; the halos hold the neighbours' boundary layers of the last frame and the
    slabs together the whole volume of that frame"""
    G = 64
    run_worker(world, "peer", tmp_path, G)
    z = [np.load(os.path.join(str(tmp_path), "halo_r%d.npz" % k)) for k in range(world)]
    for k in range(world - 1):
        up = [same_bits(z[k]["recv_hi"], z[k + 1]["hist_lo"][f]) for f in range(3)]
        down = [same_bits(z[k + 1]["recv_lo"], z[k]["hist_hi"][f]) for f in range(3)]
        assert up[2] and down[2], (k, up, down)                  # the halos hold the last frame
        assert not same_bits(z[k + 1]["hist_lo"][1], z[k + 1]["hist_lo"][2])
        assert np.nanmax(np.abs(z[k]["send_hi"])) > 0 and float(z[k]["ms"]) > 0
    capi, synth = pkg.capi, pkg.synth
    last = synth.Scene(2, 128, 106, lut_res=(32, 27, 32), seed=3, sphere_r=0.75)
    first = synth.Scene(2, 128, 106, lut_res=(32, 27, 32), seed=1, sphere_r=0.9)
    ctx = capi.Context(capi.make_config(2, (128, 106), voxel_size=2.0 / G, brick_size=8 * 2.0 / G), 0)
    inv = first.inverse((G, G, G))
    for i in range(2):
        ctx.set_calibration(i, first.xyz[i], first.lut_res, first.uv[i], first.lut_res, (0.5, 4.5))
        ctx.set_inverse_calibration(i, inv[i], (G, G, G))
    ctx.step(last.depth, last.color)
    assert same_bits(np.concatenate([zz["tsdf"] for zz in z], axis=0), ctx.readback_tsdf())
    ctx.close()


def test_halo_by_copy_engine_loopback(tmp_path):
    """the same transport with one context as its own two neighbours (slab 1 of 3; a neighbour in the caller's process is
    used without IPC): six steps, so that each staging set is refilled twice behind its readers"""
    run_worker(1, "peer_loopback", tmp_path, 96)
    z = np.load(os.path.join(str(tmp_path), "halo_r0.npz"))
    lo_matches = [same_bits(z["recv_hi"], z["hist_lo"][f]) for f in range(6)]      # lower face -> own upper halo
    hi_matches = [same_bits(z["recv_lo"], z["hist_hi"][f]) for f in range(6)]
    assert lo_matches[5] and hi_matches[5], (lo_matches, hi_matches)
    assert not same_bits(z["hist_lo"][4], z["hist_lo"][5]) and float(z["ms"]) > 0


def test_the_copy_engine_halo_reports_a_neighbour_that_stopped(pkg, monkeypatch):
    """a neighbour that does not stage its step is an error after RGBDR_PEER_TIMEOUT_S, not a hang; calls out of order are
    refused"""
    capi, synth = pkg.capi, pkg.synth
    monkeypatch.setenv("RGBDR_PEER_TIMEOUT_S", "0.5")
    G = 64
    scene = synth.Scene(2, 128, 106, lut_res=(32, 27, 32), seed=1)
    inv = scene.inverse((G, G, G))
    ctxs = []
    for r in range(2):
        c = capi.Context(capi.make_config(2, (128, 106), voxel_size=2.0 / G, brick_size=8 * 2.0 / G, slab_rank=r, slab_count=2), 0)
        for i in range(2):
            c.set_calibration(i, scene.xyz[i], scene.lut_res, scene.uv[i], scene.lut_res, (0.5, 4.5))
            c.set_inverse_calibration(i, inv[i], (G, G, G))
        ctxs.append(c)
    a, b = ctxs
    with pytest.raises(capi.RgbdrError) as e:
        a.halo_set_peer(1, b"x" * capi.HALO_PEER_BYTES)          # before its own export
    assert e.value.status == capi.ERR_STATE
    ea, eb = a.halo_export(), b.halo_export()
    with pytest.raises(capi.RgbdrError) as e:
        a.halo_set_peer(1, b"x" * capi.HALO_PEER_BYTES)          # not an export
    assert e.value.status == capi.ERR_INVALID_ARGUMENT
    with pytest.raises(capi.RgbdrError) as e:
        a.halo_set_peer(2, eb)
    assert e.value.status == capi.ERR_OUT_OF_RANGE
    a.halo_set_peer(1, eb)
    b.halo_set_peer(0, ea)
    with pytest.raises(capi.RgbdrError) as e:
        a.halo_pull_async()                                      # before begin_step + integrate
    assert e.value.status == capi.ERR_STATE
    a.halo_begin_step()
    a.step(scene.depth, scene.color)
    # b never steps.  a's side stream waits for b's faces on the device; the HOST gives the neighbour up once it is more than
    # a couple of dozen steps behind what is being enqueued, instead of queueing for ever
    steps = 0
    with pytest.raises(capi.RgbdrError) as e:
        for steps in range(1, 200):
            a.halo_pull_async()
            a.halo_begin_step()
            a.integrate()
    assert e.value.status == capi.ERR_STATE and "neighbour" in str(e.value) and 8 < steps < 64, steps
    for c in ctxs:
        c.close()                                                # (releases the waits a's streams still sit in: no hang)


def test_async_halo_exchanger_over_rccl_loopback(tmp_path):
    """the stream-ordered path as it runs on a multi-GPU node -- RCCL send/recv on the library's
    staging sets, side stream, events, no host synchronisation between four frames -- with
    the one process as its own two neighbours"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "slab_worker.py"), "loopback", str(tmp_path), "96"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    z = np.load(os.path.join(str(tmp_path), "loopback.npz"))
    lo_matches = [same_bits(z["recv_hi"], z["hist_lo"][f]) for f in range(4)]      # lower face -> own upper halo
    hi_matches = [same_bits(z["recv_lo"], z["hist_hi"][f]) for f in range(4)]
    assert lo_matches[3] and hi_matches[3], (lo_matches, hi_matches)
    assert not same_bits(z["hist_lo"][2], z["hist_lo"][3])                          # the frames differ at the faces
    assert float(z["ms"]) > 0


@pytest.mark.parametrize("world,mode", [(2, "shard"), (4, "shard"), (1, "shard_nccl"), (2, "lag"), (1, "lag_nccl"), (1, "lag_raw"),
                                        (2, "lag_pipe"), (1, "lag_raw_pipe")])
def test_sensor_sharded_chain_over_ranks(world, mode, tmp_path, pkg):
    """rgbd_recon_amd.dist.FrameGather: the pre_* chain runs for 4 / world sensors per rank, the packed frames are
    all-gathered and the brick counters all-reduced; the slabs, the occupied bricks and the composited slab ray-march of
    the last of three frames equal one unsharded context's.  (nccl: one rank on the one GPU, but the collectives run on the
    library's own buffers and stream through RCCL.)"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    G = 64
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "slab_worker.py"), mode, str(tmp_path), str(G)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    z = [np.load(os.path.join(str(tmp_path), "shard_r%d.npz" % k)) for k in range(world)]
    capi, synth = pkg.capi, pkg.synth
    n = 4
    first = synth.Scene(n, 128, 106, lut_res=(32, 27, 32), seed=1, sphere_r=0.9)
    last = synth.Scene(n, 128, 106, lut_res=(32, 27, 32), seed=3, sphere_r=0.75)
    ctx = capi.Context(capi.make_config(n, (128, 106), voxel_size=2.0 / G, brick_size=8 * 2.0 / G), 0)
    inv = first.inverse((G, G, G))
    for i in range(n):
        ctx.set_calibration(i, first.xyz[i], first.lut_res, first.uv[i], first.lut_res, (0.5, 4.5))
        ctx.set_inverse_calibration(i, inv[i], (G, G, G))
    ctx.step(last.depth, last.color)
    assert same_bits(np.concatenate([zz["tsdf"] for zz in z], axis=0), ctx.readback_tsdf())
    view = capi.make_view((2.2, 1.6, 1.9), (0.0, 0.9, 0.0), (0.0, 1.0, 0.0), 50.0, 96, 72, (-1.0, 0.0, -1.0), (1.0, 2.0, 1.0), shade_mode=0)
    view.skip_space = 1
    col, dep, _ = ctx.raymarch(view)
    for zz in z:
        assert np.array_equal(zz["counters"], ctx.readback_brick_counters())
        assert np.array_equal(zz["occupied"], ctx.get_occupied()[0])
        assert same_bits(zz["color"], col) and same_bits(zz["depth"], dep)
    assert (dep < 1).mean() > 0.02
    if mode.startswith("lag"):
        # rdist.LaggedChain: the sweep lags the chain by one frame -- after push k the volume is frame k - 1's (bit for bit: the
        # frame came through rgbdr_import_frame from the chain-only context, its gather ran on a side stream)
        mid = synth.Scene(n, 128, 106, lut_res=(32, 27, 32), seed=2, sphere_r=0.6)
        for frame, key, bricks in ((first, "tsdf_after_push_1", True), (mid, "tsdf_after_push_2", False)):
            ctx.set_use_bricks(bricks)
            ctx.step(frame.depth, frame.color)
            assert same_bits(np.concatenate([zz[key] for zz in z], axis=0), ctx.readback_tsdf()), key
    ctx.close()
