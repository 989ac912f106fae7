#!/usr/bin/env python3
"""Developer probe: does the once-per-process host stall of the lagged chain schedule (r05_notes/scaling_tail.md) also show when
the LIBRARY enqueues the gather (rgbdr_shard_allgather_async on a raw one-rank communicator: ncclAllGather + ncclAllReduce,
the calls of a real N > 1 run) instead of the send / recv pairs that stand in for the other ranks on one GPU?
Rank 1 of 4 of configs[3]; prints every push the host spent more than 5 ms in."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
load_package()
import torch
import torch.distributed as tdist
from rgbd_recon_amd import capi, synth
from rgbd_recon_amd import dist as rdist
N, K, W, H, G = 8, 4, 512, 424, 512
MODE = os.environ.get("MODE", "library")      # library | sendrecv
scene = synth.Scene(N, W, H, lut_res=(128, 106, 128), seed=1234)
dev = torch.device("cuda", 0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
tdist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
ctx = capi.Context(capi.make_config(N, (W, H), voxel_size=2.0 / G, brick_size=8 * 2.0 / G, flags=capi.FLAGS_DEFAULT,
                                    res_override=(G, G, G), slab_rank=1, slab_count=K), 0)
g = ctx.geo
chain = capi.Context(capi.make_config(N, (W, H), voxel_size=g.brick_size, brick_size=g.brick_size), 0)
for c in (ctx, chain):
    for i in range(N):
        c.set_calibration(i, scene.xyz[i], scene.lut_res, scene.uv[i], scene.lut_res, (0.5, 4.5))
for i in range(N):
    ctx.synth_inverse_calibration(i, scene.pinhole(i))
A = (torch.from_numpy(scene.depth).to(dev), torch.from_numpy(scene.color).to(dev))
torch.cuda.synchronize()
ctx.settle(3.0)
ctx.set_use_bricks(False)
ex = rdist.ManagedSlabExchange(ctx, dev, 1, K, loopback=True)          # halo faces to this GPU itself
comm = rdist.RcclComm(0, 1, None, dev)
chain.update_device(A[0].data_ptr(), A[1].data_ptr()); chain.clear_occupied_bricks(); chain.process_textures(); chain.sync()
if os.environ.get("PLAIN_FIRST"):            # the bench's order: the plain sharded schedule first, its gather on the MAIN stream
    shared = os.environ["PLAIN_FIRST"] == "same"      # ... through the SAME communicator the lagged gather uses afterwards
    comm_plain = comm if shared else rdist.RcclComm(0, 1, None, dev)
    ctx.update_device(A[0].data_ptr(), A[1].data_ptr()); ctx.clear_occupied_bricks(); ctx.process_textures(); ctx.sync()
    plain_gather = rdist.RawLoopbackGather(ctx, dev, 1, K, comm_plain)
    for k in range(110):
        ctx.update_device(A[0].data_ptr(), A[1].data_ptr()); ctx.clear_occupied_bricks(); ctx.process_textures()
        plain_gather()
        ctx.update_occupied_bricks(); ex.begin_step(); ctx.integrate(); ex.exchange_async()
    ctx.sync(); torch.cuda.synchronize(); tdist.barrier(); torch.cuda.synchronize()
    ctx.set_sensor_shard(0, 0)
if MODE == "library":
    chain.set_sensor_shard(0, N)            # (a one-rank communicator holds every sensor: the calls of a real run, one rank wide)
    lag = rdist.LaggedChain(ctx, chain, dev, None, before_sweep=ex.begin_step, after_sweep=ex.exchange_async, nccl_comm=comm.handle)
else:
    gather = rdist.RawLoopbackGather(chain, dev, 1, K, comm)
    lag = rdist.LaggedChain(ctx, chain, dev, gather, before_sweep=ex.begin_step, after_sweep=ex.exchange_async)
SYNC, BARRIER = int(os.environ.get("SYNC_EVERY", "50")), int(os.environ.get("BARRIER_EVERY", "0"))
slow, t_all = [], time.perf_counter()
for k in range(int(os.environ.get("PUSHES", "400"))):
    t0 = time.perf_counter()
    lag.push(A[0].data_ptr(), A[1].data_ptr())
    dt = time.perf_counter() - t0
    if dt > 5e-3:
        slow.append((k, round(dt * 1e3, 1)))
        if k > 0:                           # (CLOCK_MONOTONIC in us, the clock of AMD_LOG_LEVEL's lines: profiles/stall_log.sh)
            print("STALL_WINDOW %d %d" % (time.monotonic_ns() // 1000 - int(dt * 1e6), time.monotonic_ns() // 1000), flush=True)
    if SYNC and k % SYNC == SYNC - 1:       # (the bench synchronises between its phases)
        ctx.sync(); torch.cuda.synchronize()
    if BARRIER and k % BARRIER == BARRIER - 1:
        ctx.sync(); torch.cuda.synchronize(); tdist.barrier(); torch.cuda.synchronize()
ctx.sync(); torch.cuda.synchronize()
print("%s: %d pushes in %.1f ms, pushes over 5 ms: %s" % (MODE, k + 1, (time.perf_counter() - t_all) * 1e3, slow), flush=True)
lag.close(); ex.close(); comm.close(); ctx.close(); chain.close()
