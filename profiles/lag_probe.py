#!/usr/bin/env python3
"""Developer probe: in the slab timelines (r05_lag_timeline_*.txt) the sweep of the lagged schedule is 8 per cent shorter
than the one of the plain schedules.  Which difference does it?  One process, slab 1/4 of 512^3, 8 sensors, no RCCL at all:
  plain   chain and sweep in one context
  split   chain in a chain-only context, rgbdr_import_frame, sweep in the other (the lagged schedule without its gather)
with the sweep timed by HIP events around its launch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
load_package()
import torch
from rgbd_recon_amd import capi, synth
N = int(os.environ.get("SENSORS", "8")); K = int(os.environ.get("SLABS", "4"))
W, H, G = 512, 424, 512
scene = synth.Scene(N, W, H, lut_res=(128, 106, 128), seed=1234)
dev = torch.device("cuda", 0)
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
chain.set_stream(ctx.stream())
staging = os.environ.get("STAGING", "1") == "1"
if staging:
    ctx.halo_staging(1); ctx.set_halo_staging(1)
ex = None
if os.environ.get("EXCH") == "1":        # the halo exchange of the bench (RCCL to this GPU itself, on a side stream)
    import torch.distributed as tdist
    from rgbd_recon_amd import dist as rdist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    tdist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    main = torch.cuda.ExternalStream(int(ctx.stream()), device=dev)
    ex = rdist.HaloExchanger(ctx.device_tsdf(), dev, main, rank=1, world=K, ctx=ctx, loopback=True)


def plain():
    ctx.update_device(A[0].data_ptr(), A[1].data_ptr())
    ctx.clear_occupied_bricks(); ctx.process_textures(); ctx.update_occupied_bricks()
    if ex and mode["exchange"]:
        if mode.get("nowait"):
            ex.done = [None, None]
        ex.begin_step()
    if mode.get("dummy_wait"):
        e = torch.cuda.Event(); e.record(aux); main.wait_event(e)
    ctx.integrate()
    if ex and mode["exchange"]:
        ex.exchange_async()


def split_ex():
    chain.update_device(A[0].data_ptr(), A[1].data_ptr())
    chain.clear_occupied_bricks(); chain.process_textures()
    v = chain.shard_view()
    ctx.clear_occupied_bricks()
    ctx.import_frame(int(v.frames), int(v.counters))
    ctx.update_occupied_bricks()
    ex.begin_step(); ctx.integrate(); ex.exchange_async()


mode = {"exchange": False}


def split():
    chain.update_device(A[0].data_ptr(), A[1].data_ptr())
    chain.clear_occupied_bricks(); chain.process_textures()
    v = chain.shard_view()
    ctx.clear_occupied_bricks()
    ctx.import_frame(int(v.frames), int(v.counters))
    ctx.update_occupied_bricks(); ctx.integrate()


def split_lagged():
    # the order of dist.LaggedChain.push: import the frame of the step before, chain of this one, sweep
    v = chain.shard_view()
    ctx.clear_occupied_bricks()
    ctx.import_frame(int(v.frames), int(v.counters))
    chain.update_device(A[0].data_ptr(), A[1].data_ptr())
    chain.clear_occupied_bricks(); chain.process_textures()
    ctx.update_occupied_bricks(); ctx.integrate()


def run(tag, step, steps=40, warm=10):
    for _ in range(warm):
        step()
    ctx.sync(); torch.cuda.synchronize()
    ctx.set_timer_detail(0); ctx.enable_timer_accumulation(True)
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s = torch.cuda.ExternalStream(int(ctx.stream()), device=dev)
    t0.record(s)
    for _ in range(steps):
        step()
    t1.record(s)
    ctx.sync(); torch.cuda.synchronize()
    ns, n = ctx.timer_stats("2integrate")
    ctx.enable_timer_accumulation(False); ctx.enable_timers(False); ctx.set_timer_detail(2)
    print("%-62s sweep %.4f ms   frame %.4f ms" % (tag, ns / max(n, 1) * 1e-6, t0.elapsed_time(t1) / steps), flush=True)


for rep in range(2):
    run("plain", plain)
    run("split", split)
    run("split, lagged order", split_lagged)
if ex:
    aux = torch.cuda.Stream(dev)
    seq = os.environ.get("SEQ", "off,dummy,nowait,split,on").split(",")
    for rep in range(2):
        for what in seq:
            mode.update(exchange=what in ("on", "nowait"), nowait=what == "nowait", dummy_wait=what == "dummy")
            torch.cuda.synchronize()
            run({"off": "plain, exchange off", "dummy": "plain, exchange off, a cross-stream wait before the sweep",
                 "nowait": "plain + halo exchange without its wait", "split": "split + halo exchange",
                 "on": "plain + halo exchange"}[what], split_ex if what == "split" else plain)
else:
    for n in (2, 3, 4, 8, 1):
        ctx.set_sweep_launches(n)
        run("plain, sweep in %d launches" % n, plain)
ctx.close(); chain.close()
