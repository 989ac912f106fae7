"""One rank of the multi-process slab test (tests/test_dist_gpu.py); launched by
`python -m torch.distributed.run`.  All ranks share cuda:0 and talk over gloo, with
the device buffers staged through the host -- the same rgbd_recon_amd.dist functions
run over RCCL with one GPU per rank."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

BMIN, BMAX = (-1.0, 0.0, -1.0), (1.0, 2.0, 1.0)


def loopback_mode(out_dir, G):
    """One process, one GPU, real RCCL: a middle slab (rank 1 of 3) whose two neighbours are
    the process itself.  Four different frames go through integrate + the asynchronous,
    stream-ordered HaloExchanger with the library's staging sets and no host synchronisation;
    afterwards the halos must hold the own boundary layers of the LAST frame."""
    load_package()
    from rgbd_recon_amd import capi, synth
    from rgbd_recon_amd import dist as rdist

    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    scenes = [synth.Scene(2, 128, 106, lut_res=(32, 27, 32), seed=s, sphere_r=r)
              for s, r in ((1, 0.9), (2, 0.6), (3, 0.75), (4, 0.85))]
    inv = scenes[0].inverse((G, G, G))
    ctx = capi.Context(capi.make_config(2, (128, 106), voxel_size=2.0 / G, brick_size=8 * 2.0 / G, slab_rank=1, slab_count=3), 0)
    for i in range(2):
        ctx.set_calibration(i, scenes[0].xyz[i], scenes[0].lut_res, scenes[0].uv[i], scenes[0].lut_res, (0.5, 4.5))
        ctx.set_inverse_calibration(i, inv[i], (G, G, G))
    main = torch.cuda.Stream(dev)
    torch.cuda.set_stream(main)
    ctx.set_stream(main.cuda_stream)
    ex = rdist.HaloExchanger(ctx.device_tsdf(), dev, main, rank=1, world=3, ctx=ctx, loopback=True)
    frames = [(torch.from_numpy(s.depth).to(dev), torch.from_numpy(s.color).to(dev)) for s in scenes]
    lo, hi, rlo, rhi = rdist.halo_views(ctx.device_tsdf(), dev)
    torch.cuda.synchronize()
    history = []
    for n, (d, c) in enumerate(frames):
        ctx.update_device(d.data_ptr(), c.data_ptr())
        ctx.clear_occupied_bricks()
        ctx.process_textures()
        ctx.update_occupied_bricks()
        ctx.set_use_bricks(n % 2 == 0)                                # both staging paths
        ex.begin_step()
        ctx.integrate()
        ex.exchange_async()
        history.append((lo.clone(), hi.clone()))
    ex.wait()
    main.synchronize()
    np.savez(os.path.join(out_dir, "loopback.npz"), recv_lo=rlo.cpu().numpy(), recv_hi=rhi.cpu().numpy(),
             hist_lo=torch.stack([h[0] for h in history]).cpu().numpy(),
             hist_hi=torch.stack([h[1] for h in history]).cpu().numpy(), ms=ex.last_transfer_ms() or -1.0)
    dist.destroy_process_group()


def exchanger_mode(out_dir, G, library_staging=False):
    """three frames through integrate + HaloExchanger.exchange_async (gloo: staged through
    the host); afterwards every rank dumps its boundary layers and its halos"""
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    load_package()
    from rgbd_recon_amd import capi, synth
    from rgbd_recon_amd import dist as rdist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    scenes = [synth.Scene(2, 128, 106, lut_res=(32, 27, 32), seed=s, sphere_r=r) for s, r in ((1, 0.9), (2, 0.6), (3, 0.75))]
    inv = scenes[0].inverse((G, G, G))
    ctx = capi.Context(capi.make_config(2, (128, 106), voxel_size=2.0 / G, brick_size=8 * 2.0 / G, slab_rank=rank,
                                        slab_count=world), 0)
    for i in range(2):
        ctx.set_calibration(i, scenes[0].xyz[i], scenes[0].lut_res, scenes[0].uv[i], scenes[0].lut_res, (0.5, 4.5))
        ctx.set_inverse_calibration(i, inv[i], (G, G, G))
    main = torch.cuda.Stream(dev)
    torch.cuda.set_stream(main)
    ctx.set_stream(main.cuda_stream)
    ex = rdist.HaloExchanger(ctx.device_tsdf(), dev, main, rank=rank, world=world, via_host=True,
                             ctx=ctx if library_staging else None)
    frames = [(torch.from_numpy(s.depth).to(dev), torch.from_numpy(s.color).to(dev)) for s in scenes]
    torch.cuda.synchronize()
    lo, hi, rlo, rhi = rdist.halo_views(ctx.device_tsdf(), dev)
    history = []
    for n, (d, c) in enumerate(frames):
        ctx.update_device(d.data_ptr(), c.data_ptr())
        ctx.clear_occupied_bricks()
        ctx.process_textures()
        ctx.update_occupied_bricks()
        if library_staging:
            ctx.set_use_bricks(n != 1)                                # frame 1: full sweep (stages in the kernel)
            ex.begin_step()
        ctx.integrate()
        ex.exchange_async()
        history.append((lo.clone(), hi.clone()))                  # stream-ordered snapshots, no host sync
    ex.wait()
    main.synchronize()
    np.savez(os.path.join(out_dir, "halo_r%d.npz" % rank), send_lo=lo.cpu().numpy(), send_hi=hi.cpu().numpy(),
             recv_lo=rlo.cpu().numpy(), recv_hi=rhi.cpu().numpy(), tsdf=ctx.readback_tsdf(),
             hist_lo=torch.stack([h[0] for h in history]).cpu().numpy(),
             hist_hi=torch.stack([h[1] for h in history]).cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


def peer_mode(out_dir, G, loopback):
    """The halo by copy engine (rdist.PeerCopySlabExchange: rgbdr_halo_export / _set_peer / _pull_async): three frames, no host
    synchronisation in between; the ranks are PROCESSES sharing cuda:0, so the neighbours' staging sets and events are mapped
    through HIP IPC (loopback: one process, rank 1 of 3, with itself as both neighbours -- no IPC).  Dumps what
    exchanger_mode dumps."""
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    load_package()
    from rgbd_recon_amd import capi, synth
    from rgbd_recon_amd import dist as rdist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    slab_rank, slab_count = (1, 3) if loopback else (rank, world)
    dev = torch.device("cuda:0")
    scenes = [synth.Scene(2, 128, 106, lut_res=(32, 27, 32), seed=s, sphere_r=r) for s, r in ((1, 0.9), (2, 0.6), (3, 0.75))]
    inv = scenes[0].inverse((G, G, G))
    ctx = capi.Context(capi.make_config(2, (128, 106), voxel_size=2.0 / G, brick_size=8 * 2.0 / G, slab_rank=slab_rank,
                                        slab_count=slab_count), 0)
    for i in range(2):
        ctx.set_calibration(i, scenes[0].xyz[i], scenes[0].lut_res, scenes[0].uv[i], scenes[0].lut_res, (0.5, 4.5))
        ctx.set_inverse_calibration(i, inv[i], (G, G, G))
    main = torch.cuda.Stream(dev)
    torch.cuda.set_stream(main)
    ctx.set_stream(main.cuda_stream)
    ctx.enable_timers(True)
    ex = rdist.PeerCopySlabExchange(ctx, dev, slab_rank, slab_count, loopback=loopback)
    frames = [(torch.from_numpy(s.depth).to(dev), torch.from_numpy(s.color).to(dev)) for s in scenes]
    torch.cuda.synchronize()
    lo, hi, rlo, rhi = rdist.halo_views(ctx.device_tsdf(), dev)
    history = []
    order = [0, 1, 2, 0, 1, 2] if loopback else [0, 1, 2]      # (six steps: every staging set is refilled behind its readers)
    for n, k in enumerate(order):
        d, c = frames[k]
        ctx.update_device(d.data_ptr(), c.data_ptr())
        ctx.clear_occupied_bricks()
        ctx.process_textures()
        ctx.update_occupied_bricks()
        ctx.set_use_bricks(n != 1)                                    # frame 1: full sweep (stages in the kernel)
        ex.begin_step()
        ctx.integrate()
        ex.exchange_async()
        history.append((lo.clone(), hi.clone()))                  # stream-ordered snapshots, no host sync
    ex.wait()
    main.synchronize()
    ms = ex.last_transfer_ms()
    np.savez(os.path.join(out_dir, "halo_r%d.npz" % rank), send_lo=lo.cpu().numpy(), send_hi=hi.cpu().numpy(),
             recv_lo=rlo.cpu().numpy(), recv_hi=rhi.cpu().numpy(), tsdf=ctx.readback_tsdf(),
             hist_lo=torch.stack([h[0] for h in history]).cpu().numpy(),
             hist_hi=torch.stack([h[1] for h in history]).cpu().numpy(), ms=-1.0 if ms is None else ms)
    dist.barrier()
    ex.close()
    ctx.close()
    dist.destroy_process_group()


def shard_mode(out_dir, G, backend):
    """The pre_* chain sharded by sensor over the slab ranks (rdist.FrameGather): every rank runs 4 / world sensors, the
    packed frames are all-gathered and the brick counters all-reduced before updateOccupiedBricks / integrate.  Three
    different frames, both sweeps; every rank dumps its slab of the last frame, its occupied bricks and a slab
    ray-march (which shades from the gathered frames of all sensors)."""
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    load_package()
    from rgbd_recon_amd import capi, synth
    from rgbd_recon_amd import dist as rdist

    dev = torch.device("cuda:0")
    if backend == "nccl":
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 4
    scenes = [synth.Scene(n, 128, 106, lut_res=(32, 27, 32), seed=s, sphere_r=r) for s, r in ((1, 0.9), (2, 0.6), (3, 0.75))]
    inv = scenes[0].inverse((G, G, G))
    ctx = capi.Context(capi.make_config(n, (128, 106), voxel_size=2.0 / G, brick_size=8 * 2.0 / G, slab_rank=rank,
                                        slab_count=world), 0)
    for i in range(n):
        ctx.set_calibration(i, scenes[0].xyz[i], scenes[0].lut_res, scenes[0].uv[i], scenes[0].lut_res, (0.5, 4.5))
        ctx.set_inverse_calibration(i, inv[i], (G, G, G))
    gather = rdist.FrameGather(ctx, dev, rank=rank, world=world, via_host=backend != "nccl")
    assert (gather.first, gather.count) == (rank * (n // world), n // world)
    frames = [(torch.from_numpy(s.depth).to(dev), torch.from_numpy(s.color).to(dev)) for s in scenes]
    torch.cuda.synchronize()
    for k, (d, c) in enumerate(frames):
        ctx.set_pipelined(k == 1)                                     # one frame in the two-stream schedule
        ctx.set_use_bricks(k != 1)
        ctx.update_device(d.data_ptr(), c.data_ptr())
        ctx.clear_occupied_bricks()
        ctx.process_textures()
        gather()
        ctx.update_occupied_bricks()
        ctx.integrate()
    ctx.sync()
    if world > 1:
        rdist.exchange_halo_via_host(rdist.halo_views(ctx.device_tsdf(), dev), rank=rank, world=world)
    torch.cuda.synchronize()
    view = capi.make_view((2.2, 1.6, 1.9), (0.0, 0.9, 0.0), (0.0, 1.0, 0.0), 50.0, 96, 72, BMIN, BMAX, shade_mode=0)
    view.skip_space = 1
    col, dep, ns = rdist.raymarch_slabs(ctx, view, dev, via_host=backend != "nccl")
    np.savez(os.path.join(out_dir, "shard_r%d.npz" % rank), tsdf=ctx.readback_tsdf(), occupied=ctx.get_occupied()[0],
             counters=ctx.readback_brick_counters(), color=col.cpu().numpy(), depth=dep.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


def lag_mode(out_dir, G, backend, pipelined=False):
    """rdist.LaggedChain: a chain-only context runs frame k+1 (sharded by sensor) before the slab context sweeps frame k;
    the gather of frame k+1 runs on a side stream under that sweep and the slab context takes the completed frame with
    rgbdr_import_frame.  Three different frames: after the third push the volume is the SECOND frame's, after flush()
    the third's; both sweeps.  Dumps what shard_mode dumps, plus the volume before the flush."""
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    load_package()
    from rgbd_recon_amd import capi, synth
    from rgbd_recon_amd import dist as rdist

    dev = torch.device("cuda:0")
    if backend in ("nccl", "raw"):
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 4
    scenes = [synth.Scene(n, 128, 106, lut_res=(32, 27, 32), seed=s, sphere_r=r) for s, r in ((1, 0.9), (2, 0.6), (3, 0.75))]
    inv = scenes[0].inverse((G, G, G))
    ctx = capi.Context(capi.make_config(n, (128, 106), voxel_size=2.0 / G, brick_size=8 * 2.0 / G, slab_rank=rank,
                                        slab_count=world), 0)
    # the chain-only context: same sensors, box and brick size (hence the same brick grid), a token volume of one voxel per brick
    chain = capi.Context(capi.make_config(n, (128, 106), voxel_size=8 * 2.0 / G, brick_size=8 * 2.0 / G), 0)
    assert tuple(chain.geo.res_bricks) == tuple(ctx.geo.res_bricks)
    for i in range(n):
        for c in (ctx, chain):
            c.set_calibration(i, scenes[0].xyz[i], scenes[0].lut_res, scenes[0].uv[i], scenes[0].lut_res, (0.5, 4.5))
        ctx.set_inverse_calibration(i, inv[i], (G, G, G))
    raw = None
    if backend == "raw":       # the library enqueues the gather itself on a raw communicator (the C ABI's managed form)
        raw = rdist.RcclComm(rank, world, None, dev)
        chain.set_sensor_shard(rank * (n // world), n // world)
        lag = rdist.LaggedChain(ctx, chain, dev, None, nccl_comm=raw.handle)
    else:
        gather = rdist.FrameGather(chain, dev, rank=rank, world=world, via_host=backend != "nccl") if world > 1 or backend == "nccl" else None
        lag = rdist.LaggedChain(ctx, chain, dev, gather)
    frames = [(torch.from_numpy(s.depth).to(dev), torch.from_numpy(s.color).to(dev)) for s in scenes]
    torch.cuda.synchronize()
    before = []
    if pipelined:
        # the SWEEPING context on its two-stream schedule: its import copies run on its second stream while the chain context
        # already works on the next frame on another one (rgbdr_import_frame_from orders both directions itself).  Twelve
        # frames without a host synchronisation, so that a copy overtaken by the next chain would show as a torn frame.
        ctx.set_pipelined(True)
        for k in range(9):
            d, c = frames[k % 3]
            ctx.set_use_bricks(k % 2 == 0)
            lag.push(d.data_ptr(), c.data_ptr())
        ctx.sync()
    for k, (d, c) in enumerate(frames):
        ctx.set_use_bricks(k != 2)            # the sweep of frame k-1 happens in push k: frame 0 bricked, frame 1 full
        lag.push(d.data_ptr(), c.data_ptr())
        ctx.sync()
        before.append(ctx.readback_tsdf() if k > 0 else None)
    ctx.set_use_bricks(True)
    lag.flush()
    ctx.sync()
    ctx.update_device(frames[2][0].data_ptr(), frames[2][1].data_ptr())     # (the colour frame the shading of the ray-march reads)
    if world > 1:
        rdist.exchange_halo_via_host(rdist.halo_views(ctx.device_tsdf(), dev), rank=rank, world=world)
    torch.cuda.synchronize()
    view = capi.make_view((2.2, 1.6, 1.9), (0.0, 0.9, 0.0), (0.0, 1.0, 0.0), 50.0, 96, 72, BMIN, BMAX, shade_mode=0)
    view.skip_space = 1
    col, dep, ns = rdist.raymarch_slabs(ctx, view, dev, via_host=backend == "gloo")
    np.savez(os.path.join(out_dir, "shard_r%d.npz" % rank), tsdf=ctx.readback_tsdf(), occupied=ctx.get_occupied()[0],
             counters=ctx.readback_brick_counters(), color=col.cpu().numpy(), depth=dep.cpu().numpy(),
             tsdf_after_push_1=before[1], tsdf_after_push_2=before[2])
    if raw is not None:
        ctx.sync(); chain.sync()
        raw.close()
    dist.barrier()
    dist.destroy_process_group()


def main():
    if sys.argv[1] in ("lag", "lag_nccl", "lag_raw", "lag_pipe", "lag_raw_pipe"):
        return lag_mode(sys.argv[2], int(sys.argv[3]), {"lag": "gloo", "lag_nccl": "nccl", "lag_raw": "raw", "lag_pipe": "gloo",
                                                         "lag_raw_pipe": "raw"}[sys.argv[1]], pipelined=sys.argv[1].endswith("_pipe"))
    if sys.argv[1] in ("shard", "shard_nccl"):
        return shard_mode(sys.argv[2], int(sys.argv[3]), "nccl" if sys.argv[1] == "shard_nccl" else "gloo")
    if sys.argv[1] == "loopback":
        return loopback_mode(sys.argv[2], int(sys.argv[3]))
    if sys.argv[1] in ("peer", "peer_loopback"):
        return peer_mode(sys.argv[2], int(sys.argv[3]), loopback=sys.argv[1] == "peer_loopback")
    if sys.argv[1] in ("exchanger", "exchanger_lib"):
        return exchanger_mode(sys.argv[2], int(sys.argv[3]), library_staging=sys.argv[1] == "exchanger_lib")
    out_dir, G, limit = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
    backend = sys.argv[4] if len(sys.argv) > 4 else "gloo"
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    load_package()
    from rgbd_recon_amd import capi, synth
    from rgbd_recon_amd import dist as rdist

    dev = torch.device("cuda:0")
    if backend == "nccl":
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = synth.Scene(2, 128, 106, lut_res=(32, 27, 32))
    inv = scene.inverse((G, G, G))

    def make(**slab):
        ctx = capi.Context(capi.make_config(2, (128, 106), voxel_size=2.0 / G, brick_size=8 * 2.0 / G, tsdf_limit=limit, **slab), 0)
        for i in range(2):
            ctx.set_calibration(i, scene.xyz[i], scene.lut_res, scene.uv[i], scene.lut_res, (0.5, 4.5))
            ctx.set_inverse_calibration(i, inv[i], (G, G, G))
        ctx.step(scene.depth, scene.color)
        ctx.sync()
        return ctx

    ctx = make(slab_rank=rank, slab_count=world)
    if world > 1:
        rdist.exchange_halo_via_host(rdist.halo_views(ctx.device_tsdf(), dev), rank=rank, world=world)
    torch.cuda.synchronize()
    whole = make() if rank == 0 else None
    for n, (shade_mode, eye, skip) in enumerate([(0, (2.2, 1.6, 1.9), 0), (1, (0.85, 1.7, 0.8), 1)]):
        view = capi.make_view(eye, (0.0, 0.9, 0.0), (0.0, 1.0, 0.0), 50.0, 96, 72, BMIN, BMAX, shade_mode=shade_mode)
        view.skip_space = skip
        col, dep, ns = rdist.raymarch_slabs(ctx, view, dev, via_host=backend != "nccl")
        np.savez(os.path.join(out_dir, "slab_r%d_v%d.npz" % (rank, n)), color=col.cpu().numpy(), depth=dep.cpu().numpy(),
                 ns=ns.cpu().numpy())
        if whole is not None:
            c, d, s = whole.raymarch(view)
            np.savez(os.path.join(out_dir, "whole_v%d.npz" % n), color=c, depth=d, ns=s)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
