"""N > 1 path on CPU: two gloo ranks each own a Z slab (rgbdr_slab_range through
the C ABI), fill it from the oracle, run the SAME halo-exchange function the GPU
path uses (rgbd-recon_amd/dist.py) and check the halos against the neighbour's
boundary tile layers."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def tile_layers(vol, t0, t1):
    """[Z,Y,X] volume -> tile-linear layers [t1-t0, TY*TX*512] (layout of include/rgbdr.h)"""
    Z, Y, X = vol.shape
    ty, tx = Y // 8, X // 8
    v = vol[t0 * 8:t1 * 8].reshape(t1 - t0, 8, ty, 8, tx, 8)       # tz, z, ty, y, tx, x
    return np.ascontiguousarray(v.transpose(0, 2, 4, 1, 3, 5)).reshape(t1 - t0, -1)


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from __graft_entry__ import load_oracle, load_package

    load_package()
    from rgbd_recon_amd import capi, synth
    from rgbd_recon_amd import dist as rdist

    orc = load_oracle()
    orc.set_threads(2)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    G = 32
    scene = synth.Scene(2, 48, 40, lut_res=(12, 10, 12), seed=99)
    inv = scene.inverse((G, G, G))
    cfg = capi.make_config(2, (48, 40), voxel_size=2.0 / G, brick_size=8 * 2.0 / G, slab_rank=rank, slab_count=world)
    g = capi.compute_geometry(cfg)
    # every rank runs the (cheap) pre_* chain redundantly; integration only on its slab
    ref = orc.run_pipeline(scene, synth.BBOX_MIN, synth.BBOX_MAX, (G, G, G), None, brick_size=g.brick_size,
                           bv=g.brick_voxels, res_bricks=tuple(g.res_bricks))
    vol = orc.integrate(inv, ref["sil"], ref["depth_b"], ref["quality"], (G, G, G), 0.01, ref["mask"],
                        res_bricks=tuple(g.res_bricks), z_range=(g.slab_voxel_z0, g.slab_voxel_z1),
                        bbox=(synth.BBOX_MIN, synth.BBOX_MAX), brick_size=g.brick_size)
    own = tile_layers(vol, g.slab_tile_z0, g.slab_tile_z1)
    layer = own.shape[1]
    slab = torch.full((own.shape[0] + 2, layer), float("nan"))       # [halo_lo | owned | halo_hi]
    slab[1:-1] = torch.from_numpy(own)
    rdist.exchange_halo(slab[1], slab[-2], slab[0], slab[-1], rank=rank, world=world)
    np.save(os.path.join(tmp, "slab%d.npy" % rank), slab.numpy())
    np.save(os.path.join(tmp, "range%d.npy" % rank), np.array([g.slab_tile_z0, g.slab_tile_z1]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_slab_halo_exchange_gloo(world, tmp_path, orc, pkg):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    # reference: the whole volume on one rank
    G = 32
    scene = pkg.synth.Scene(2, 48, 40, lut_res=(12, 10, 12), seed=99)
    inv = scene.inverse((G, G, G))
    g = pkg.capi.compute_geometry(pkg.capi.make_config(2, (48, 40), voxel_size=2.0 / G, brick_size=8 * 2.0 / G))
    ref = orc.run_pipeline(scene, pkg.synth.BBOX_MIN, pkg.synth.BBOX_MAX, (G, G, G), inv, brick_size=g.brick_size,
                           bv=g.brick_voxels, res_bricks=tuple(g.res_bricks))
    full = tile_layers(ref["tsdf"], 0, g.tiles[2])
    covered = 0
    for r in range(world):
        slab = np.load(os.path.join(str(tmp_path), "slab%d.npy" % r))
        t0, t1 = np.load(os.path.join(str(tmp_path), "range%d.npy" % r))
        assert np.array_equal(slab[1:-1], full[t0:t1], equal_nan=True)          # slabs tile the volume exactly
        if r > 0:
            assert np.array_equal(slab[0], full[t0 - 1], equal_nan=True)        # lower halo = neighbour's top layer
        else:
            assert np.all(np.isnan(slab[0]))                                      # outer faces stay untouched
        if r < world - 1:
            assert np.array_equal(slab[-1], full[t1], equal_nan=True)
        else:
            assert np.all(np.isnan(slab[-1]))
        covered += t1 - t0
    assert covered == g.tiles[2]


def _composite_worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    from __graft_entry__ import load_package

    load_package()
    from rgbd_recon_amd import dist as rdist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(5)
    h, w = 9, 13
    owner = rng.integers(-1, world, size=(h, w))                     # -1: no slab hit the pixel
    truth = rng.standard_normal((h, w, 4)).astype(np.float32)
    truth[0, 0] = [-0.0, np.nan, np.inf, -1.0]                       # bit patterns a float sum would lose
    owner[0, 0] = world - 1
    tdepth = rng.random((h, w)).astype(np.float32)
    mine = torch.from_numpy(owner == rank)
    color = np.tile(np.float32([0, 1, 0, 0]), (h, w, 1))
    depth = np.ones((h, w), np.float32)
    color[owner == rank], depth[owner == rank] = truth[owner == rank], tdepth[owner == rank]
    col, dep = rdist.composite_slab_frames(torch.from_numpy(color), torch.from_numpy(depth), mine)
    np.savez(os.path.join(tmp, "comp%d.npz" % rank), col=col.numpy(), dep=dep.numpy(), owner=owner, truth=truth, tdepth=tdepth)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_slab_frame_compositing_gloo(world, tmp_path, pkg):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_composite_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), "comp%d.npz" % r))
        hit = z["owner"] >= 0
        want_c = np.where(hit[..., None], z["truth"], np.float32([0, 1, 0, 0]))
        want_d = np.where(hit, z["tdepth"], np.float32(1))
        assert want_c.view(np.uint32).tolist() == z["col"].view(np.uint32).tolist()
        assert want_d.view(np.uint32).tolist() == z["dep"].view(np.uint32).tolist()
