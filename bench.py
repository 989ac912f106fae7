#!/usr/bin/env python3
"""bench.py -- TSDF fusion + depth preprocessing on MI355X.

A step is one pass of the hot path over one synthetic frame set already resident
in HBM: NetKinectArray::update (device->device copy of the resident frames),
clearOccupiedBricks, processTextures (morph/bilateral/boundary/normal/quality),
updateOccupiedBricks and a FULL-SWEEP integrate() of every voxel
(source/kinect_client.cpp:572-602).  The brick-skipping mode the reference
defaults to is timed separately and reported under "bricked".

N = 1: BASELINE.json configs[2] "4 sensors, 512^3 TSDF, full pre_* chain".
N > 1: one process per GPU, the volume is split into Z slabs of storage-tile
layers (no data-path collective for integration; the one exchange per step is the
halo tile layers to the Z neighbours over RCCL):
  --gpus 2 / 4: BASELINE.json configs[3], "8 sensors, 512^3 TSDF, Z-slab split" (strong split
                of the same volume; configs[3] names 4 GPUs, 2 is the same workload on 2);
  --gpus 8:     BASELINE.json configs[4], "8 sensors, 1024^3 TSDF across 8 MI355X +
                tsdf_colorfill/inpaint post-pass" (the slab ray-march + hole filling is timed
                and reported under "post_pass", outside `value`);
  --weak:       the weak-scaling grids instead (4 sensors; 512^3 / 512x512x1024 / 512x1024x1024 /
                1024^3 for 1 / 2 / 4 / 8 GPUs, 134 M voxels per GPU).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_oracle, load_package  # noqa: E402

# weak scaling: ~134 M voxels (one 512^3 worth) per GPU over the same 2 m box; every
# axis stays a power of two so the 1:1 inverse LUT needs no interpolation
GRID_FOR_GPUS = {1: (512, 512, 512), 2: (512, 512, 1024), 4: (512, 1024, 1024), 8: (1024, 1024, 1024)}
HBM_PEAK = 8.0e12  # B/s, MI355X_MICROARCH.md "HBM3E peak BW"


def choose_workload(world, loopback=False, weak=False, sensors=0, cubic_grid=0):
    """Which BASELINE.json config a run with `world` GPUs is: (sensors, grid, config string, scaling)."""
    if world == 1 or loopback or weak:
        n = sensors or 4
        grid = GRID_FOR_GPUS.get(world, (512, 512, 512))
        cfg = "configs[2]: 4 sensors, 512^3 TSDF, full pre_* depth-filter chain on 1 MI355X" if world == 1 and not loopback \
            else ("configs[2] at fixed work per GPU (weak scaling): %d sensors, 512^3 voxels per MI355X as Z slabs of a %dx%dx%d TSDF, "
                  "staged RCCL halo exchange per step" % ((n,) + tuple(grid)))
        scaling = "weak"
    elif world == 8:
        n = sensors or 8
        grid = (1024, 1024, 1024)
        cfg = "configs[4]: 8 sensors, 1024^3 TSDF across 8 MI355X + tsdf_colorfill/inpaint post-pass"
        scaling = "weak"                 # 134 M voxels per GPU, like the 512^3 of one GPU
    else:
        n = sensors or 8
        grid = (512, 512, 512)
        cfg = "configs[3]: 8 sensors, 512^3 TSDF, Z-slab split across %d MI355X with RCCL brick-halo over xGMI" % world
        scaling = "strong"
    if cubic_grid:
        grid = (cubic_grid,) * 3
    return n, grid, cfg, scaling


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--grid", type=int, default=0, help="override with a cubic grid of this size")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (gloo: debugging several "
                                                      "ranks on one GPU)")
    ap.add_argument("--sensors", type=int, default=0, help="0 = what the BASELINE config of --gpus names (4 at 1 GPU, 8 above)")
    ap.add_argument("--weak", action="store_true", help="N > 1: only the weak-scaling grids (4 sensors, 512^3 voxels per GPU), without the "
                                                        "BASELINE configs[3]/[4] run that the default adds under baseline_configs_run")
    ap.add_argument("--baseline-configs", action="store_true",
                    help="N > 1: make BASELINE configs[3] (N = 2, 4) / configs[4] (N = 8) -- 8 sensors -- the headline and report the "
                         "weak-scaling twin under weak_scaling_4_sensors (the default is the other way round: the headline of an N > 1 "
                         "run is the N = 1 workload at fixed work per GPU, so that value(N) is comparable with N x value(1))")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pipeline", action="store_true",
                    help="RGBDR_FLAG_PIPELINE for the headline: the pre_* chain of step k+1 overlaps integrate of step k on a "
                         "second stream (2-3 %% more frames/s, but the integrate launches it is measured on run 4 %% longer "
                         "under the overlap; the other schedule is always reported under 'other_schedule')")
    ap.add_argument("--loopback", action="store_true",
                    help="one GPU, real RCCL: run an inner Z slab (rank 1 of 4) whose two neighbours are this process "
                         "itself -- exercises the whole N > 1 code path (probe, staging, side stream); value is per slab")
    ap.add_argument("--arena-trials", type=int, default=0,
                    help="RGBDR_ARENA_TRIALS for the headline context; 0 (default) = leave the library's own default in "
                         "force (up to 16 candidate placements of the LUT arena for arenas of 1 GiB and more; an RGBDR_ARENA_TRIALS "
                         "already in the environment is honoured).  Within one box the sweep time differs by up to 12 %% "
                         "with where hipMalloc placed the arena; the candidates' times are reported in "
                         "roofline.arena_placement_probe_ms, next to what the first placement alone (frac_first_placement) and "
                         "round 3's library default of three (frac_first_3) would have given")
    ap.add_argument("--cpu-rows", type=int, default=0,
                    help="bound the CPU baseline to this many z rows of the volume (0 = the whole volume, about 10 s)")
    ap.add_argument("--slab", default="",
                    help="r/k: one GPU, real RCCL: run Z slab r of k of BASELINE configs[3] (k = 2, 4: 8 sensors, 512^3) or "
                         "configs[4] (k = 8: 8 sensors, 1024^3) exactly as rank r of a --gpus k run would -- the sweep that "
                         "stages its boundary layers, the exchange on the side stream with this process as its own "
                         "neighbour(s); value is this slab's share")
    ap.add_argument("--slab-sweep", type=int, default=0,
                    help="k: --slab r/k for every r in one process, one JSON line with a row per rank (projection of the "
                         "multi-GPU balance from single-GPU runs; DESIGN.md section 6)")
    ap.add_argument("--no-shard", dest="shard", action="store_false",
                    help="N > 1: every rank runs the pre_* chain for ALL sensors (rounds 1-3) instead of its n / N share followed by "
                         "the all-gather of the packed frames and the all-reduce of the brick counters (rgbdr_set_sensor_shard; "
                         "measured per rank of configs[3]: 0.68 against 0.71-0.73 ms per frame, profiles/r04_notes)")
    ap.add_argument("--torch-collectives", dest="managed", action="store_false",
                    help="N > 1 over RCCL: halo exchange and frame gather through torch.distributed's process group (rounds 1-3) "
                         "instead of the C ABI's managed forms, where the LIBRARY enqueues them on its own streams with a raw RCCL "
                         "communicator (what a C++ host does, host/slab_loop.cpp); gloo runs always go through torch")
    ap.set_defaults(shard=True, managed=True)
    ap.add_argument("--launch-timeout", type=float, default=3600.0, help="--gpus N > 1 self-launch: seconds before the ranks are stopped")
    args = ap.parse_args()

    # `python3 bench.py --gpus N` without a launcher around it (how the driver starts it): this process becomes
    # the parent of N ranks.  It must not touch the GPU (nor import torch) before or after that.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:], timeout=args.launch_timeout))
    # stdout carries the one JSON line and nothing else: RCCL's version banner, gloo's connection notes and any
    # other chatter of the libraries below go to stderr
    global EMIT
    sys.stdout.flush()
    EMIT = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    if args.slab_sweep:
        sys.exit(slab_sweep(args))
    # What an N > 1 run times.  Default: the N = 1 workload at fixed work per GPU -- 4 sensors, 512^3 voxels per
    # rank as Z slabs of one larger volume, halo exchange and all ("scaling": "weak") -- as the headline, so that value(N)
    # compares with N x value(1); BASELINE.json's own multi-GPU configs (8 sensors: configs[3] at N = 2 / 4, configs[4] at
    # N = 8) are timed in the same run and reported under baseline_configs_run.  --baseline-configs swaps the roles,
    # --weak drops the second run.
    args.twin = None
    if args.gpus > 1 and not args.sensors and not args.grid:
        if args.baseline_configs:
            args.weak, args.twin = False, "weak"
        elif args.weak:
            args.twin = None
        else:
            args.weak, args.twin = True, "baseline"
    run_rank(args)


EMIT = sys.stdout


def emit(obj):
    EMIT.write(json.dumps(obj) + "\n")
    EMIT.flush()


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n, argv, timeout=3600.0, child_cmd=None, poll_s=0.2):
    """Parent of a `--gpus n` run that was started as a plain process: one child per rank with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in its environment (what torch.distributed.run would set; the
    reference's frame loop is one process too, source/kinect_client.cpp:1013-1014).  The parent never touches
    the GPU, so starting children is not an exec from a GPU process.  Rank 0 inherits stdout and prints the
    one JSON line; the other ranks' stdout goes to stderr.  Returns 0 when every rank did; otherwise the
    first failing rank's code after stopping the rest (by their own PIDs), 124 after `timeout`."""
    import subprocess
    cmd = list(child_cmd) if child_cmd else [sys.executable, os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env.setdefault("MASTER_PORT", str(free_port()))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update({"WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n)})
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen(cmd, env=e, stdout=None if r == 0 else sys.stderr))
    deadline = time.monotonic() + timeout
    rc = 0
    live = set(range(n))
    while live and rc == 0:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0:
                rc = code if code > 0 else 128 - code
                sys.stderr.write("[bench launcher] rank %d exited with status %d; stopping the other ranks\n" % (r, code))
                break
        if live and rc == 0:
            if time.monotonic() > deadline:
                rc = 124
                sys.stderr.write("[bench launcher] %d rank(s) still running after %.0f s; stopping them\n" % (len(live), timeout))
                break
            time.sleep(poll_s)
    if rc != 0:
        for r in live:
            procs[r].terminate()
        t_kill = time.monotonic() + 10.0
        for r in live:
            try:
                procs[r].wait(max(0.1, t_kill - time.monotonic()))
            except subprocess.TimeoutExpired:
                procs[r].kill()
                procs[r].wait()
    return rc


def parse_slab(text):
    """'r/k' -> (r, k)"""
    try:
        r, k = (int(t) for t in text.split("/"))
    except ValueError:
        raise SystemExit("--slab takes r/k, e.g. 1/4")
    if k not in (2, 4, 8) or not 0 <= r < k:
        raise SystemExit("--slab r/k: k is 2 or 4 (configs[3]) or 8 (configs[4]), 0 <= r < k")
    return r, k


def run_rank(args, slab=None, quiet=False, shared=None):
    """One rank of the benchmark: the whole single-GPU run, rank `RANK` of a --gpus N run, or (slab = (r, k))
    slab r of k on this GPU with itself as its neighbours.  Returns the JSON object (printed by rank 0 unless quiet)."""
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and not (world == 1 and args.gpus <= 1):
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    if slab is None and args.slab:
        slab = parse_slab(args.slab)
    if slab is None and args.loopback:
        slab = (1, 4)                    # --loopback: an inner slab of configs[3]
    if slab is not None and world != 1:
        raise SystemExit("--slab / --loopback run on one GPU (they stand in for a --gpus k run)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the hot path has no CPU fallback")
    if args.backend == "gloo":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    loop = slab is not None              # one GPU stands in for rank r of k: it is its own neighbour(s)
    multi = world > 1 or loop
    shared = shared if shared is not None else {}
    if loop:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
    if multi and not shared.get("pg"):
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
        shared["pg"] = True

    # several ranks on one GPU (--backend gloo, debugging): no placement shopping, it would hold world x 10 arenas
    if args.backend == "gloo" and world > 1:
        os.environ["RGBDR_ARENA_TRIALS"] = "1"
    elif args.arena_trials > 0:
        os.environ["RGBDR_ARENA_TRIALS"] = str(args.arena_trials)     # read by the library when the LUT arena is created
    trials = os.environ.get("RGBDR_ARENA_TRIALS", "library default (16)")
    load_package()
    from rgbd_recon_amd import capi, synth
    from rgbd_recon_amd import dist as rdist

    W, H = 512, 424
    slab_rank, slab_count = slab if loop else (rank, world)
    N, grid, baseline_config, scaling = choose_workload(slab_count, False, args.weak, args.sensors, args.grid)
    G = grid[0]
    if shared.get("scene_n") != N:
        shared["scene"], shared["scene_n"] = synth.Scene(N, W, H, lut_res=(128, 106, 128), seed=1234), N
        shared["d_depth"] = torch.from_numpy(shared["scene"].depth).to(dev)
        shared["d_color"] = torch.from_numpy(shared["scene"].color).to(dev)
    scene, d_depth, d_color = shared["scene"], shared["d_depth"], shared["d_color"]
    flags = capi.FLAGS_DEFAULT | (capi.FLAG_PIPELINE if args.pipeline else 0)
    cfg = capi.make_config(N, (W, H), voxel_size=2.0 / G, brick_size=8 * 2.0 / G, flags=flags,
                           res_override=grid, slab_rank=slab_rank, slab_count=slab_count)
    ctx = capi.Context(cfg, local_rank)
    g = ctx.geo
    for i in range(N):
        ctx.set_calibration(i, scene.xyz[i], scene.lut_res, scene.uv[i], scene.lut_res, (0.5, 4.5))
        ctx.synth_inverse_calibration(i, scene.pinhole(i))
    torch.cuda.synchronize()

    halo, transport = None, None
    if multi:
        halo = rdist.halo_views(ctx.device_tsdf(), dev)
        # The library enqueues on a torch stream so that the RCCL exchange can be ordered
        # against the kernels with events instead of host syncs (rgbd_recon_amd.dist.HaloExchanger:
        # boundary layers are staged device-to-device, the transfer of step k overlaps step k+1).
        main = torch.cuda.Stream(dev)
        torch.cuda.set_stream(main)
        if not (args.managed and args.backend == "nccl"):
            ctx.set_stream(main.cuda_stream)
        # Probe the device transport once before anything is timed.  If RCCL point-to-point
        # on these buffers fails on this node, say so in the JSON line and stop: a run whose halos
        # go through host memory would measure PCIe, not xGMI.
        transport = {"kind": "rccl" if args.backend == "nccl" else args.backend + " (host-staged)", "group": None}
        if args.backend == "nccl":
            if "fallback" not in shared:
                shared["fallback"] = dist.new_group(backend="gloo")
            ok, why = 1, ""
            try:
                ctx.sync()
                rdist.exchange_halo(*halo, rank=slab_rank, world=slab_count, loopback=loop)
                torch.cuda.synchronize()
            except Exception as e:  # noqa: BLE001 -- reported, not swallowed
                ok, why = 0, "%s: %s" % (type(e).__name__, str(e).splitlines()[0][:200])
            flag = torch.tensor([ok], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=shared["fallback"])
            if int(flag[0]) == 0:
                # (--backend gloo asks for the host-staged path explicitly, for debugging several ranks on one GPU)
                sys.stderr.write("[bench rank %d] RCCL point-to-point failed (%s); refusing to time a host-staged "
                                 "fallback\n" % (rank, why or "on another rank"))
                if rank == 0:
                    emit({"error": "RCCL point-to-point halo exchange failed: %s" % (why or "on another rank"), "n_gpus": world})
                ctx.close()
                dist.destroy_process_group()
                sys.exit(3)
        if not (args.managed and args.backend == "nccl"):
            exchanger = rdist.HaloExchanger(ctx.device_tsdf(), dev, main, rank=slab_rank, world=slab_count, group=transport["group"],
                                            via_host=transport["kind"] != "rccl", ctx=ctx, loopback=loop)
    # The pre_* chain sharded by sensor over the ranks: rank r runs it for N / k sensors, the packed frames are all-gathered
    # and the brick counters all-reduced on the chain's stream (SURVEY 8e's alternative to the redundant chain; the chain's
    # time then shrinks with the number of GPUs like the sweep's).  On one GPU standing in for a rank (--slab / --loopback)
    # the other ranks' sensors come from two unsharded frames of the same static scene and the gather's traffic is
    # reproduced by RCCL send / recv to this process itself (dist.FrameGather loopback).
    gather = None
    want_shard = multi and args.shard and N % slab_count == 0 and N > 1
    managed = bool(multi and args.managed and transport["kind"] == "rccl")
    if multi and (want_shard or managed):
        if loop:
            for _ in range(2):                      # both frame buffers of the two-stream schedule hold every sensor's frame
                ctx.update_device(d_depth.data_ptr(), d_color.data_ptr())
                ctx.clear_occupied_bricks(); ctx.process_textures(); ctx.update_occupied_bricks(); ctx.integrate()
            ctx.sync()
        if managed:
            made = None
            try:
                ctx.enable_timers(True)
                if os.environ.get("RGBDR_BENCH_FAIL_MANAGED") == "construct":   # test hook: tests/test_bench_gpu.py walks the fallback roads
                    raise RuntimeError("RGBDR_BENCH_FAIL_MANAGED=construct")
                made = rdist.ManagedSlabExchange(ctx, dev, slab_rank, slab_count, group=transport["group"], shard=want_shard, loopback=loop)
            except Exception as e:  # noqa: BLE001 -- a raw communicator that does not come up must not cost the run
                sys.stderr.write("[bench rank %d] library-managed RCCL unavailable (%s: %s)\n" % (rank, type(e).__name__, str(e)[:200]))
            ok = 1 if made is not None else 0
            if world > 1:   # every rank takes the same road: one rank without its communicator sends all of them to torch.distributed
                flag = torch.tensor([ok], dtype=torch.int32)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=shared["fallback"])
                ok = int(flag[0])
            if ok:
                exchanger = made
                gather = exchanger.gather if exchanger.shard else None
            else:
                sys.stderr.write("[bench rank %d] using torch.distributed for the exchange\n" % rank)
                if made is not None:
                    try:
                        made.close()
                    except Exception:  # noqa: BLE001
                        pass
                managed = False
                ctx.set_sensor_shard(0, 0)
                ctx.set_stream(main.cuda_stream)
                exchanger = rdist.HaloExchanger(ctx.device_tsdf(), dev, main, rank=slab_rank, world=slab_count, group=transport["group"],
                                                via_host=False, ctx=ctx, loopback=loop)
        if not managed and want_shard:
            gather = rdist.FrameGather(ctx, dev, rank=slab_rank, world=slab_count, group=transport["group"],
                                       via_host=transport["kind"] != "rccl", loopback=loop)

    def step(bricks):
        ctx.update_device(d_depth.data_ptr(), d_color.data_ptr())
        ctx.clear_occupied_bricks()
        ctx.process_textures()
        if gather is not None:
            gather()                         # the other ranks' sensors: all-gather of the packed frames, all-reduce of the brick counts
        ctx.update_occupied_bricks()
        if halo is not None:
            exchanger.begin_step()           # the sweep stores its boundary layers into a staging set
        ctx.integrate()
        if halo is not None:
            exchanger.exchange_async()

    def barrier():
        ctx.sync()
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(bricks, steps, warmup, detail=0, sample_box=None):
        ctx.set_use_bricks(bricks)
        for _ in range(warmup):
            step(bricks)
        barrier()
        # timed region: only the integrate launches carry HIP events (their duration is needed
        # for the roofline; each event record costs ~4 us of stream time); the per-pass breakdown
        # comes from a separate short run
        ctx.set_timer_detail(detail)
        ctx.enable_timer_accumulation(True)
        t0 = time.perf_counter()
        for _ in range(steps):
            step(bricks)
        timed.host_enqueue_ms = (time.perf_counter() - t0) / steps * 1e3   # the host's share: enqueue time per step
        if sample_box is not None:       # the queue still holds most of the steps: clocks / power under load
            sample_box.update(gpu_state())
        barrier()
        dt = time.perf_counter() - t0
        names = ("2integrate",) + (("1preprocess", "bricks") if detail > 0 else ()) + \
                (("morph", "bilateral", "boundary", "normal", "quality") if detail > 1 else ())
        stats = {n: ctx.timer_stats(n) for n in names}
        ctx.enable_timer_accumulation(False)
        ctx.enable_timers(False)
        ctx.set_timer_detail(2)          # the library's default again (detail 0 mutes every timer but "2integrate")
        timed.local_dt = dt
        if multi:
            t = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, stats

    # Untimed set-up: memory a previous process released is wiped by the driver in the background
    # for a while (a 6 GB free slows the sweep by 4 % for ~0.2 s).  Wait until the
    # sweep time has settled before the warm-up and the timed steps begin.
    if managed:
        # The library-managed exchange has never run between two devices (the pool has one GPU per box): its first step is a
        # trial.  If it fails on ANY rank, every rank goes back to torch.distributed for the exchange and to the redundant
        # chain, and the line says so -- a scaling run must not be lost to it.
        ok = 1
        try:
            step(False)
            exchanger.wait()
            ctx.sync()
            if os.environ.get("RGBDR_BENCH_FAIL_MANAGED") == "trial":
                raise RuntimeError("RGBDR_BENCH_FAIL_MANAGED=trial")
        except Exception as e:  # noqa: BLE001
            ok = 0
            sys.stderr.write("[bench rank %d] library-managed exchange failed in its trial step (%s: %s)\n" % (rank, type(e).__name__, str(e)[:200]))
        if world > 1:
            flag = torch.tensor([ok], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=shared["fallback"])
            ok = int(flag[0])
        if not ok:
            managed = False
            try:
                exchanger.close()
            except Exception:  # noqa: BLE001
                pass
            gather = None
            ctx.set_sensor_shard(0, 0)
            ctx.set_halo_staging(-1)
            ctx.set_stream(main.cuda_stream)
            exchanger = rdist.HaloExchanger(ctx.device_tsdf(), dev, main, rank=slab_rank, world=slab_count, group=transport["group"],
                                            via_host=False, ctx=ctx, loopback=loop)
    step(False)
    ctx.settle(3.0)
    if multi:
        dist.barrier()

    # ---- headline: full sweep ------------------------------------------------
    box = {}
    dt, stats = timed(False, args.steps, args.warmup, sample_box=box)
    V_local = g.res_volume[0] * g.res_volume[1] * (g.slab_voxel_z1 - g.slab_voxel_z0)
    V_total = g.res_volume[0] * g.res_volume[1] * g.res_volume[2] if not loop else V_local   # loopback: this slab only
    ms_per_step = dt / args.steps * 1e3
    value = V_total / (dt / args.steps) / 1e6
    int_ns, int_n = stats["2integrate"]
    int_s = int_ns / max(int_n, 1) * 1e-9
    # algorithmic bytes of one integrate launch on this rank (DESIGN.md "Algorithmic bytes"):
    # one f32 store per voxel + the three f32 LUT planes per voxel and sensor (the
    # repacked, xyz-only 1:1 LUT) + the packed 8-B frame texels read once
    bytes_launch = V_local * (4 + 12 * N) + N * W * H * 8
    achieved = bytes_launch / int_s if int_s > 0 else 0.0
    # Box calibration, right after the timed steps: one replay of the kernel's own memory streams with no
    # arithmetic (k_arena_probe: non-temporal 16-B loads of every LUT plane of the kept arena + the
    # non-temporal TSDF tile stores, same block -> tile order).  MI355X boxes of the pool stream this at
    # 5.9-6.7 TB/s depending on the box; kernel rate / replay rate says how close the kernel is to what THIS
    # box moves, whatever its level.
    replay_ms = ctx.settle(0.0)
    box_stream = V_local * (4 + 12 * N) / (replay_ms * 1e-3) if replay_ms > 0 else 0.0
    local_ms_per_step = timed.local_dt / args.steps * 1e3
    host_enqueue_ms = timed.host_enqueue_ms
    halo_ms = exchanger.last_transfer_ms() if multi else None
    plain_ms = None
    if loop:
        # what the staging costs the sweep: the same slab without a staging set (plain kernel, no exchange)
        halo_keep, halo = halo, None
        ctx.set_halo_staging(-1)
        dt_plain, stats_plain = timed(False, args.steps, args.warmup)
        halo = halo_keep
        plain_ms = (stats_plain["2integrate"][0] / max(stats_plain["2integrate"][1], 1) * 1e-6, dt_plain / args.steps * 1e3)
    # every rank's numbers on rank 0: the N > 1 line carries per-rank arrays and prices the SLOWEST rank's kernel
    # (rank 0 is an edge slab with one neighbour and one staged face; inner slabs stage two)
    per_rank = None
    if world > 1:
        mine = torch.tensor([int_s * 1e3, -1.0 if halo_ms is None else halo_ms, local_ms_per_step, float(bytes_launch),
                             replay_ms], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        rows = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(rows, mine)
        rows = [r.cpu().tolist() for r in rows]
        slowest = max(range(world), key=lambda r: rows[r][0])
        per_rank = {"integrate_ms": [round(r[0], 4) for r in rows],
                    "halo_ms": [None if r[1] < 0 else round(r[1], 4) for r in rows],
                    "ms_per_step": [round(r[2], 4) for r in rows],
                    "roofline_frac": [round(r[3] / (r[0] * 1e-3) / HBM_PEAK, 4) if r[0] > 0 else None for r in rows],
                    "box_stream_replay_ms": [round(r[4], 4) for r in rows],
                    "slowest_rank": slowest}
        int_s, bytes_launch, replay_ms = rows[slowest][0] * 1e-3, int(rows[slowest][3]), rows[slowest][4]
        achieved = bytes_launch / int_s if int_s > 0 else 0.0
        box_stream = (bytes_launch - N * W * H * 8) / (replay_ms * 1e-3) if replay_ms > 0 else 0.0

    # breakdown, not part of the headline timing: the totals from a run with the three total timers,
    # the five passes from a run with every timer (their event records inflate the totals)
    lean = bool(shared.get("lean"))      # --slab-sweep: only the headline of every slab
    if not lean:
        _, tot_stats = timed(False, 5, 1, detail=1)
        _, pass_stats = timed(False, 5, 1, detail=2)
        stats.update({k: v for k, v in tot_stats.items() if k not in stats})
        stats.update({k: v for k, v in pass_stats.items() if k not in stats})

    # ---- brick-skipping mode (reference default) -------------------------------
    bsteps = max(args.steps // 2, 1)
    dtb, stats_b = timed(True, bsteps, 2) if not lean else (0.0, {"2integrate": (0, 0)})
    occ = ctx.occupied_ratio()
    bint_ns, bint_n = stats_b["2integrate"]

    # brick-skipping mode with the pre_* chain of frame k+1 overlapping the sweep of frame k (RGBDR_FLAG_PIPELINE):
    # the sweep is short here, so the two streams overlap for most of it
    bricked_pipelined = None
    if world == 1 and not loop and not args.pipeline:
        ctx.set_pipelined(True)
        dtbp, _ = timed(True, bsteps, 2)
        ctx.set_pipelined(False)
        bricked_pipelined = round(dtbp / bsteps * 1e3, 4)

    # ---- the other schedule (extra keys): whichever of sequential / pipelined the headline did not use ----
    other = None
    if world == 1 and not loop:
        ctx.set_pipelined(not args.pipeline)
        dto, stats_o = timed(False, args.steps, args.warmup)
        ctx.set_pipelined(bool(args.pipeline))
        oi_ns, oi_n = stats_o["2integrate"]
        other = {"schedule": "sequential" if args.pipeline else "pipelined (pre_* of step k+1 on a second stream under integrate of step k)",
                 "ms_per_step": round(dto / args.steps * 1e3, 4), "value": round(V_total / (dto / args.steps) / 1e6, 1),
                 "integrate_ms": round(oi_ns / max(oi_n, 1) * 1e-6, 4)}

    # ---- RGBDR_FLAG_ELIDE_STORES (extra keys): the full sweep without re-storing tiles that stay -limit ----
    elided = None
    if world == 1 and not loop:
        ctx.set_elide_stores(True)
        dte, stats_e = timed(False, args.steps, args.warmup)
        ctx.set_elide_stores(False)
        ei_ns, ei_n = stats_e["2integrate"]
        elided = {"ms_per_step": round(dte / args.steps * 1e3, 4), "value": round(V_total / (dte / args.steps) / 1e6, 1),
                  "integrate_ms": round(ei_ns / max(ei_n, 1) * 1e-6, 4)}

    # ---- RGBDR_FLAG_SKIP_BACKGROUND (extra keys): LUT planes of (tile, sensor) pairs whose frame window decides the
    # outcome stay unread, tiles that are constants are not rewritten while they hold their constant ----
    skipbg = None
    if world == 1 and not loop:
        ctx.set_skip_background(True)
        dts, stats_s = timed(False, args.steps, args.warmup)
        skipped, total = ctx.skipped_pairs()
        verdicts = ctx.readback_skip_tables(0)
        ctx.set_skip_background(False)
        si_ns, si_n = stats_s["2integrate"]
        listed = int((verdicts == 0).any(axis=1).sum())
        # bytes a steady-state sweep asks for: per pair the four words the classifier reads; per listed tile its
        # list entry, its TSDF store and the LUT planes of its undecided sensors; the frame texels (windows) once
        nbytes = int(total * 16 + listed * (8 + 2048) + (total - skipped) * 3 * 512 * 4 + N * W * H * 8)
        skipbg = {"ms_per_step": round(dts / args.steps * 1e3, 4), "value": round(V_total / (dts / args.steps) / 1e6, 1),
                  "integrate_ms": round(si_ns / max(si_n, 1) * 1e-6, 4),
                  "pairs_decided": int(skipped), "pairs": int(total), "frac_decided": round(skipped / max(total, 1), 4),
                  "verdicts": {k: int((verdicts == i).sum()) for i, k in enumerate(("none", "carve", "in_front", "hidden"))},
                  "tiles_listed": listed, "tiles": int(verdicts.shape[0]),
                  "bytes_per_launch": nbytes, "GBps": round(nbytes / (si_ns / max(si_n, 1)), 1)}

    out = {
        "metric": "Mvoxels/s TSDF integration (%d sensors, %s grid) + frames/s" % (
            N, "%d^3" % grid[0] if grid[0] == grid[1] == grid[2] else "%dx%dx%d" % grid),
        "value": round(value, 1),
        "unit": "Mvoxels/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "frames_per_s": round(args.steps / dt, 2),
        "host_enqueue_ms_per_step": round(host_enqueue_ms, 4),   # when this approaches ms_per_step the host loop is the limit
        "higher_is_better": True,
        "scaling": scaling if world > 1 else None,      # one GPU: nothing scales
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "voxel_sensor_updates_per_s": round(V_total * N / (dt / args.steps), 1),
        "config": {"workload": "%d sensors 512x424 -> %dx%dx%d TSDF, full pre_* chain + full-sweep integrate, 1:1 "
                               "inverse LUT" % ((N,) + tuple(g.res_volume)),
                   "baseline_config": baseline_config,
                   "grid": list(g.res_volume), "sensors": N, "tsdf_limit": 0.01,
                   "schedule": "pipelined (pre_* of step k+1 overlaps integrate of step k)" if args.pipeline else "sequential",
                   "parallelism": ("zslab%d" % world if world > 1 else "single") + (
                       " (loopback: slab %d of %d on one GPU, its own neighbour over RCCL)" % (slab_rank, slab_count) if loop else ""),
                   "halo_transport": transport["kind"] if multi else None,
                   "pre_chain": ("sharded by sensor: %d of %d sensors per rank, packed frames all-gathered + brick counters all-reduced on "
                                 "the chain's stream" % (N // slab_count, N)) if gather is not None else "every sensor on every rank",
                   "collectives": (("library-managed RCCL (C ABI: rgbdr_halo_exchange_async, rgbdr_shard_allgather)" if managed
                                    else "torch.distributed") if multi else None)},
        "roofline": {"bound": "hbm", "kernel": "rgbdr::k_integrate_tiled<%d, 4, true, false, %s>" % (N, "true" if multi else "false"),
                     "achieved": round(achieved / 1e9, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK, 4), "traffic": None, "traffic_source": None,
                     "bytes_per_launch": bytes_launch, "avg_launch_ms": round(int_s * 1e3, 4),
                     "launches_timed": int_n, "rank": per_rank["slowest_rank"] if per_rank else (slab_rank if loop else 0),
                     "box_stream_GBps": round(box_stream / 1e9, 1), "box_stream_replay_ms": round(replay_ms, 4),
                     "frac_of_box_stream": round(achieved / box_stream, 4) if box_stream > 0 else None,
                     "box": box,
                     "arena_placement_probe_ms": ctx.arena_probe()[0], "arena_kept": ctx.arena_probe()[1],
                     "arena_trials_used": trials},
        "passes_ms": {k: round(v[0] / max(v[1], 1) * 1e-6, 4) for k, v in stats.items()},
        "bricked": None if lean else {"ms_per_step": round(dtb / bsteps * 1e3, 4),
                                      "value": round(V_total / (dtb / bsteps) / 1e6, 1),
                                      "integrate_ms": round(bint_ns / max(bint_n, 1) * 1e-6, 4),
                                      "occupied_ratio": round(occ, 4), "ms_per_step_pipelined": bricked_pipelined},
        "other_schedule": other,
        "full_sweep_store_elision": elided,
        "full_sweep_background_skip": skipbg,
    }
    # What the first placement gives (RGBDR_ARENA_TRIALS=1: no probing) and what the library's default gives (unset: the
    # best of the first three candidates): the stream replay of those candidates, priced like the kernel (which runs at
    # frac_of_box_stream of its replay).
    probe_ms, kept = ctx.arena_probe()
    if len(probe_ms) > 1 and probe_ms[0] > 0 and probe_ms[kept] > 0 and world == 1:
        first_ms = int_s * 1e3 * probe_ms[0] / probe_ms[kept]
        out["roofline"]["avg_launch_ms_first_placement"] = round(first_ms, 4)
        out["roofline"]["frac_first_placement"] = round(bytes_launch / (first_ms * 1e-3) / HBM_PEAK, 4)
        # (until round 4 bench.py probed more placements than the library's default and this key scaled the result back;
        # now the headline context runs on the library's default, so it is `frac` itself unless --arena-trials was given)
        dflt_ms = int_s * 1e3 * min(m for m in probe_ms[:16] if m > 0) / probe_ms[kept]
        out["roofline"]["frac_library_default"] = round(bytes_launch / (dflt_ms * 1e-3) / HBM_PEAK, 4)
        first3_ms = int_s * 1e3 * min(m for m in probe_ms[:3] if m > 0) / probe_ms[kept]
        out["roofline"]["frac_first_3"] = round(bytes_launch / (first3_ms * 1e-3) / HBM_PEAK, 4)   # round 3's library default
        if args.arena_trials == 0 and "RGBDR_ARENA_TRIALS" not in os.environ:
            out["roofline"]["frac_best_of_16"] = out["roofline"]["frac"]   # the library's default IS up to 16 candidates now
        out["roofline"]["placement_note"] = ("RGBDR_ARENA_TRIALS = %s; %d placements were probed: `frac` is on the one the library "
                                             "kept, frac_first_placement / frac_library_default scale the measured launch time by "
                                             "replay(candidate 0) / replay(kept) and by replay(best of the candidates) / replay(kept), "
                                             "frac_first_3 by replay(best of the first three) / replay(kept)" % (trials, len(probe_ms)))
    elif world == 1:
        out["roofline"]["frac_first_placement"] = out["roofline"]["frac"]       # a single placement was looked at
        out["roofline"]["frac_library_default"] = out["roofline"]["frac"]
        out["roofline"]["avg_launch_ms_first_placement"] = out["roofline"]["avg_launch_ms"]
    if per_rank is not None:
        out["per_rank"] = per_rank
        # BASELINE.json's multi-GPU configs name 8 sensors, its single-GPU config 4: a voxel of the N > 1 runs costs
        # twice the LUT bytes of a voxel of the N = 1 run, so `value` (Mvoxels/s) is not comparable across that step
        out["scaling_note"] = (
            "weak scaling of the N = 1 workload: every GPU owns 512^3 voxels of a %dx%dx%d volume and sweeps them from 4 sensors, "
            "value(N) compares with N x value(1); BASELINE.json's multi-GPU configs (8 sensors) are under baseline_configs_run" % tuple(grid)
            if args.weak else
            "N = 1 runs configs[2] (4 sensors), N = 2 / 4 configs[3] (8 sensors, 512^3), N = 8 configs[4] (8 sensors, 1024^3): compare "
            "voxel_sensor_updates_per_s across N, not value; the fixed-work-per-GPU twin is under weak_scaling_4_sensors")
    if loop:
        out["slab"] = {"rank": slab_rank, "of": slab_count, "owned_z_rows": int(g.slab_voxel_z1 - g.slab_voxel_z0),
                       "faces_staged": int(slab_rank > 0) + int(slab_rank < slab_count - 1),
                       "integrate_ms": round(int_s * 1e3, 4), "integrate_ms_without_staging": round(plain_ms[0], 4),
                       "staging_overhead_ms": round(int_s * 1e3 - plain_ms[0], 4),
                       "ms_per_step": round(ms_per_step, 4), "ms_per_step_without_halo": round(plain_ms[1], 4),
                       "host_enqueue_ms_per_step": round(host_enqueue_ms, 4),
                       "roofline_frac": round(achieved / HBM_PEAK, 4), "halo_ms_to_self": halo_ms,
                       "frame_gather_ms_to_self": gather.last_ms() if hasattr(gather, "last_ms") else None,
                       "schedule": ("pipelined" if args.pipeline else "sequential") + (", sharded chain" if gather is not None else "") +
                                   (", library-managed RCCL" if managed else "") + (
                                       ", RGBDR_CU_SPLIT=" + os.environ["RGBDR_CU_SPLIT"] if os.environ.get("RGBDR_CU_SPLIT") else "")}
    traffic_file = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(traffic_file):
        try:
            t = json.load(open(traffic_file))
            key = "%dx%d" % (N, G)
            if key in t and world == 1 and not loop and tuple(g.res_volume) == (G, G, G):
                # NOT measured in this run: the PMC passes of profiles/collect_pmc.sh on the same kernel and workload
                out["roofline"]["traffic"] = t[key]["hbm_bytes_per_launch"]
                out["roofline"]["traffic_source"] = "profiles/traffic.json (rocprofv3 --pmc passes of %s, not this run)" % t[key].get(
                    "source", "profiles/collect_pmc.sh")
        except Exception:
            pass

    # ---- consumer of the volume (BASELINE config 5 names the post-pass): extra keys ----
    if world == 1 and not loop:
        try:
            ctx.set_use_bricks(False)
            ctx.integrate()
            ctx.set_timer_detail(2)
            ctx.enable_timers(True)
            view = capi.make_view((2.4, 1.8, 2.1), (0.0, 0.7, 0.0), (0.0, 1.0, 0.0), 45.0, 1280, 720, synth.BBOX_MIN,
                                  synth.BBOX_MAX)
            ctx.raymarch(view)
            _, depth_img, _ = ctx.raymarch(view)
            ctx.fill_colors(1280, 720)
            ctx.fill_colors(1280, 720)
            full_ms = ctx.timer_ns("draw") * 1e-6
            view.skip_space = 1                 # brick depth peels -> start positions (reference default)
            ctx.raymarch(view)
            ctx.raymarch(view)
            out["post_pass"] = {"viewport": [1280, 720], "raymarch_ms": round(full_ms, 4),
                                "raymarch_skip_space_ms": round(ctx.timer_ns("draw") * 1e-6, 4),
                                "brickdraw_ms": round(ctx.timer_ns("brickdraw") * 1e-6, 4),
                                "holefill_ms": round(ctx.timer_ns("holefill") * 1e-6, 4),
                                "surface_pixels": round(float((depth_img < 1).mean()), 4)}
            ctx.enable_timers(False)
        except capi.RgbdrError as e:           # never let the extra keys break the headline
            out["post_pass"] = {"error": str(e)}

    # ---- the same step fed from HOST buffers (never part of `value`): extra keys ----
    if world == 1 and not loop:
        try:
            def fed(upload, steps=40):
                ctx.set_use_bricks(False)
                for _ in range(3):
                    upload()
                    ctx.clear_occupied_bricks(); ctx.process_textures(); ctx.update_occupied_bricks(); ctx.integrate()
                ctx.sync()
                t0 = time.perf_counter()
                for _ in range(steps):
                    upload()
                    ctx.clear_occupied_bricks(); ctx.process_textures(); ctx.update_occupied_bricks(); ctx.integrate()
                ctx.sync()
                return (time.perf_counter() - t0) / steps * 1e3

            depth_h, color_h = np.ascontiguousarray(scene.depth), np.ascontiguousarray(scene.color)
            pageable = fed(lambda: ctx.update(depth_h, color_h))

            def mapped_fill():                                  # the producer memcpys into the page-locked back buffer
                d, c = ctx.map_frame_buffer()
                d[:] = depth_h.view(np.uint8).reshape(-1)
                c[:] = color_h.reshape(-1)
                ctx.upload_mapped_frame()

            def mapped_only():                                  # the producer filled it on its own thread
                ctx.map_frame_buffer()
                ctx.upload_mapped_frame()

            mapped_fill(); mapped_fill()
            out["host_fed"] = {"bytes_per_frame": int(depth_h.nbytes + color_h.nbytes),
                               "ms_per_step_pageable_upload": round(pageable, 4),
                               "ms_per_step_mapped_buffer_incl_fill": round(fed(mapped_fill), 4),
                               "ms_per_step_mapped_buffer": round(fed(mapped_only), 4)}
            if not args.pipeline:       # RGBDR_FLAG_PIPELINE: upload + pre_* of frame k+1 overlap integrate of frame k
                ctx.set_pipelined(True)
                out["host_fed"]["ms_per_step_mapped_buffer_pipelined"] = round(fed(mapped_only), 4)
                ctx.set_pipelined(False)
        except capi.RgbdrError as e:
            out["host_fed"] = {"error": str(e)}

    # ---- the reference's own default operating point (extra keys) ----------------------------
    # voxel 0.01 m over (-1,0,-1)-(1,2.2,1) -> 200 x 221 x 200, bricks of 0.1 m (10 voxels), inverse LUTs at
    # the calib_inverter default spacing 0.007 m (286 x 315 x 286, generated on the device, resampled to the
    # grid at upload), DXT1 colour frames, 1280 x 1080 colour next to 512 x 424 depth, brick-skipping sweep
    if world == 1 and not loop:
        try:
            bmax = (1.0, 2.2, 1.0)
            sc = synth.Scene(N, W, H, lut_res=(128, 106, 128), seed=1234, color_wh=(1280, 1080))
            rc = capi.Context(capi.make_config(N, (W, H), color_wh=(1280, 1080), bbox_max=bmax, voxel_size=0.01, brick_size=0.1,
                                               compress_rgb=1), local_rank)
            t0 = time.perf_counter()
            for i in range(N):
                rc.set_calibration(i, sc.xyz[i], sc.lut_res, sc.uv[i], sc.lut_res, (0.5, 4.5))
                rc.set_inverse_calibration(i, rc.generate_inverse_lut(i, (286, 315, 286)), (286, 315, 286))
            rc.sync()
            t_lut = time.perf_counter() - t0
            blocks = np.stack([synth.encode_dxt(sc.color[i], 1) for i in range(N)])
            d_b = torch.from_numpy(np.ascontiguousarray(blocks)).to(dev)
            d_d = torch.from_numpy(sc.depth).to(dev)
            torch.cuda.synchronize()

            def rstep():
                rc.update_device(d_d.data_ptr(), d_b.data_ptr())
                rc.clear_occupied_bricks(); rc.process_textures(); rc.update_occupied_bricks(); rc.integrate()
            for _ in range(5):
                rstep()
            rc.sync()
            t0 = time.perf_counter()
            for _ in range(100):
                rstep()
            rc.sync()
            ms = (time.perf_counter() - t0) / 100 * 1e3
            out["reference_defaults"] = {"grid": list(rc.geo.res_volume), "brick_voxels": int(rc.geo.brick_voxels),
                                         "inverse_lut": [286, 315, 286], "colour": "DXT1 1280x1080",
                                         "ms_per_frame": round(ms, 4), "frames_per_s": round(1e3 / ms, 1),
                                         "occupied_ratio": round(rc.occupied_ratio(), 4),
                                         "inverse_luts_generated_and_resampled_s": round(t_lut, 3)}
            rc.close()
        except (capi.RgbdrError, TypeError, ValueError) as e:
            out["reference_defaults"] = {"error": str(e)}

    # ---- CPU baseline: the oracle, bounded sample, rank 0 at N=1 only ---------
    if world == 1 and not loop and rank == 0 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(ctx, scene, capi, synth, G, N, W, H, args.cpu_rows, V_total)

    if multi:
        torch.cuda.synchronize()
        out["halo"] = {"layers_per_face": int(g.halo_tile_layers), "bytes_per_face": int(halo[0].numel() * 4),
                       "transfer_ms_rank0": exchanger.last_transfer_ms(),
                       "transfer_ms_max": max([h for h in per_rank["halo_ms"] if h is not None], default=None) if per_rank else halo_ms}
    # ---- post-pass across the slabs (BASELINE configs[4]): slab ray-march (find, all-reduce MIN, shade,
    # composite) + tsdf_inpaint / tsdf_colorfill of the composited frame; outside `value` ----
    if multi and not lean:                # --loopback / --slab run it too (one slab's share of the frame): the same code path
        try:
            ctx.set_use_bricks(False)
            step(False)
            exchanger.wait()
            barrier()
            view = capi.make_view((2.4, 1.8, 2.1), (0.0, 0.7, 0.0), (0.0, 1.0, 0.0), 45.0, 1280, 720, synth.BBOX_MIN,
                                  synth.BBOX_MAX)
            vh = transport["kind"] != "rccl"
            rdist.raymarch_slabs(ctx, view, dev, group=transport["group"], via_host=vh)
            barrier()
            t0 = time.perf_counter()
            for _ in range(3):
                col, dep, _ = rdist.raymarch_slabs(ctx, view, dev, group=transport["group"], via_host=vh)
            barrier()
            t_march = (time.perf_counter() - t0) / 3 * 1e3
            ctx.set_timer_detail(2)
            ctx.enable_timers(True)
            ctx.upload_view_frame(col.cpu().numpy(), dep.cpu().numpy())
            ctx.fill_colors(1280, 720)
            ctx.fill_colors(1280, 720)
            out["post_pass"] = {"viewport": [1280, 720], "slab_raymarch_composited_ms": round(t_march, 4),
                                "holefill_ms": round(ctx.timer_ns("holefill") * 1e-6, 4),
                                "surface_pixels": round(float((dep < 1).float().mean()), 4)}
            ctx.enable_timers(False)
        except Exception as e:  # noqa: BLE001 -- extra keys must never cost the headline line
            out["post_pass"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
    if multi and hasattr(exchanger, "close"):
        ctx.sync()
        exchanger.close()                 # the raw RCCL communicators of the library-managed exchange
    ctx.close()
    if multi:
        torch.cuda.set_stream(torch.cuda.default_stream(dev))
    # The same job at fixed work per GPU (extra key): BASELINE.json's metric names 4 sensors into 512^3 on one GPU, its
    # multi-GPU configs 8 sensors -- so next to configs[3] / configs[4] the run also times the weak-scaling grid with the
    # N = 1 sensor count (134 M voxels and 4 sensors per GPU: 512x512x1024 / 512x1024x1024 / 1024^3), whose value is
    # directly comparable with N times the N = 1 value.
    twin = getattr(args, "twin", None)
    if world > 1 and not loop and twin and not lean:
        key = "weak_scaling_4_sensors" if twin == "weak" else "baseline_configs_run"
        try:
            a2 = argparse.Namespace(**vars(args))
            a2.weak, a2.twin = twin == "weak", None
            sub = dict(shared)
            sub.update(lean=True, keep_pg=True)
            w = run_rank(a2, quiet=True, shared=sub)
            for k in ("scene", "scene_n", "d_depth", "d_color"):
                shared.pop(k, None)
            out[key] = {"baseline_config": w["config"]["baseline_config"], "grid": w["config"]["grid"], "sensors": w["config"]["sensors"],
                        "scaling": w["scaling"], "value": w["value"], "ms_per_step": w["ms_per_step"], "frames_per_s": w["frames_per_s"],
                        "voxel_sensor_updates_per_s": w["voxel_sensor_updates_per_s"],
                        "per_rank": w.get("per_rank"), "roofline_frac_slowest_rank": w["roofline"]["frac"],
                        "pre_chain": w["config"]["pre_chain"], "collectives": w["config"]["collectives"],
                        "comparable_with": ("N x the value of the N = 1 run (same sensors, same voxels per GPU)" if twin == "weak" else
                                            "the 1-GPU time of the same config (8 sensors: DESIGN.md 6 has the denominators); across "
                                            "N by voxel_sensor_updates_per_s, not by value")}
        except Exception as e:  # noqa: BLE001 -- an extra key must never cost the line
            out[key] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
    if rank == 0 and not quiet:
        emit(out)
    if multi and not shared.get("keep_pg"):
        dist.destroy_process_group()
        shared["pg"] = False
    return out


def slab_sweep(args):
    """--slab-sweep k: every slab r of k of BASELINE configs[3] (k = 2, 4) / configs[4] (k = 8) on this one GPU, one
    after the other in this process, each exactly as rank r of a --gpus k run executes it (sweep with staging,
    exchange over RCCL to itself on the side stream).  One JSON line: a row per rank, the spread, and the frame rate
    a k-GPU run would show if nothing but the slowest rank's step bounded it.  A projection from single-GPU runs,
    not a scaling measurement."""
    k = args.slab_sweep
    parse_slab("0/%d" % k)
    shared = {"lean": True, "keep_pg": True}
    rows, line = [], None
    for r in range(k):
        line = run_rank(args, slab=(r, k), quiet=True, shared=shared)
        row = dict(line["slab"])
        row["arena_placement_probe_ms"] = line["roofline"]["arena_placement_probe_ms"]
        row["frac_of_box_stream"] = line["roofline"]["frac_of_box_stream"]
        rows.append(row)
        sys.stderr.write("[slab %d/%d] %s\n" % (r, k, json.dumps(row)))
    import torch.distributed as dist
    dist.destroy_process_group()
    steps = [r["ms_per_step"] for r in rows]
    ints = [r["integrate_ms"] for r in rows]
    grid = line["config"]["grid"]
    V = grid[0] * grid[1] * grid[2]
    out = {"metric": line["metric"], "unit": "Mvoxels/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "projection": "single-GPU per-rank runs of %s -- no scaling curve was measured" % line["config"]["baseline_config"],
           "ranks": rows,
           "integrate_ms_max": max(ints), "integrate_ms_min": min(ints), "integrate_spread": round(max(ints) / min(ints), 4),
           "ms_per_step_max": max(steps), "ms_per_step_min": min(steps),
           "projected_value_if_bound_by_slowest_rank": round(V / (max(steps) * 1e-3) / 1e6, 1),
           "config": line["config"], "dtype": "f32", "data": "synthetic"}
    emit(out)
    return 0


def gpu_state():
    """power / clocks of the first GPU right after the timed steps (a box that shows hundreds of watts with
    this process idle is shared).  Read from sysfs: no child process, because under
    `rocprofv3 --pmc` every child inherits the profiler's preloaded library and a script child (rocm-smi is
    one) re-executes itself after that library has initialised the GPU.  rocm-smi is the fallback only when
    sysfs has nothing and no profiler is preloaded."""
    import glob
    keep = {}
    for dev in sorted(glob.glob("/sys/class/drm/card*/device")):
        if not os.path.exists(dev + "/pp_dpm_mclk"):
            continue
        for name in ("sclk", "mclk", "fclk"):
            try:
                cur = [ln for ln in open(f"{dev}/pp_dpm_{name}").read().splitlines() if ln.rstrip().endswith("*")]
                if cur:
                    keep[name] = cur[0].split(":", 1)[1].strip(" *")
            except OSError:
                pass
        try:
            keep["perf_level"] = open(dev + "/power_dpm_force_performance_level").read().strip()
        except OSError:
            pass
        for pw in glob.glob(dev + "/hwmon/hwmon*/power1_average") + glob.glob(dev + "/hwmon/hwmon*/power1_input"):
            try:
                keep["power_W"] = round(int(open(pw).read()) / 1e6, 1)
                break
            except (OSError, ValueError):
                pass
        if keep:
            return keep
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or os.environ.get("LD_PRELOAD"):
        return {"unavailable": "profiler preloaded"}
    import subprocess
    try:
        r = subprocess.run(["rocm-smi", "-d", "0", "--showpower", "--showclocks", "--showperflevel", "--json"],
                           capture_output=True, text=True, timeout=30)
        card = next(iter(json.loads(r.stdout).values()))
        for k, v in card.items():
            kl = k.lower()
            if "power" in kl or kl.startswith("sclk") or kl.startswith("mclk") or kl.startswith("fclk") or "performance" in kl:
                keep[k] = v
        return keep
    except Exception as e:  # noqa: BLE001 -- diagnostic only
        return {"unavailable": type(e).__name__}


def available_cpus():
    """CPUs this process may actually use: the affinity mask capped by the cgroup CPU
    quota (the GPU box reports 256 logical CPUs but grants 16 CPUs of time; running
    256 OpenMP threads there is 20x slower than 16-32)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return None


def cpu_baseline(ctx, scene, capi, synth, G, N, W, H, rows, V_total):
    """Times the CPU oracle ("port") on this box's host cores over the benchmark
    workload itself: the whole pre_* chain for the frame set plus integrate() of the
    volume, taken in chunks of 64 z rows (the 1:1 LUT rows of a chunk are read back
    from the device first, untimed).  `rows` > 0 bounds the sample to that many rows
    and extrapolates.  Every chunk is also compared with the HIP TSDF bit for bit,
    so a default bench run is a full-volume parity check at the benchmark size."""
    orc = load_oracle()
    cores = available_cpus()
    threads = orc.set_threads(cores)
    g = ctx.geo
    Z = g.res_volume[2]
    ctx.set_use_bricks(False)
    ctx.integrate()
    hip = ctx.readback_tsdf()
    sil = [ctx.readback_image(capi.IMG_SILHOUETTE, i) for i in range(N)]
    db = [ctx.readback_image(capi.IMG_DEPTH_B_RG, i) for i in range(N)]
    q = [ctx.readback_image(capi.IMG_QUALITY, i) for i in range(N)]
    total_rows = Z if rows <= 0 else max(8, min(rows, Z))
    chunk = 64
    z_first = 0 if total_rows == Z else (Z // 2 // 8) * 8          # a bounded sample is taken mid-volume
    total_rows = min(total_rows, Z - z_first)
    t_int, parity, done = 0.0, True, 0
    warm = True
    for z0 in range(z_first, z_first + total_rows, chunk):
        n = min(chunk, z_first + total_rows - z0)
        inv = [ctx.readback_inverse_calibration(i, z0, z0 + n) for i in range(N)]
        if warm:                                                     # page in the library and the thread pool
            orc.integrate(inv, sil, db, q, (g.res_volume[0], g.res_volume[1], n), 0.01)
            warm = False
        t0 = time.perf_counter()
        ref = orc.integrate(inv, sil, db, q, (g.res_volume[0], g.res_volume[1], n), 0.01)
        t_int += time.perf_counter() - t0
        got = hip[z0:z0 + n]
        parity = parity and bool(np.all((ref == got) | (np.isnan(ref) & np.isnan(got))))
        done += n
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        orc.run_pipeline(scene, synth.BBOX_MIN, synth.BBOX_MAX, tuple(g.res_volume), None, brick_size=g.brick_size,
                         bv=tuple(g.brick_voxels_axis), res_bricks=tuple(g.res_bricks))
        times.append(time.perf_counter() - t0)
    t_pre = sorted(times)[1]
    t_full = t_pre + t_int * (Z / done)
    try:   # the reference's own per-resize / per-frame CPU work on this path, single-threaded as in the reference
        ref_cpu = orc.reference_cpu_work(synth.BBOX_MIN, synth.BBOX_MAX, tuple(g.res_volume), g.brick_size,
                                         ctx.readback_brick_counters(), 10)
    except MemoryError:
        ref_cpu = None
    ref_text = reference_text_baseline(ctx, scene, capi, synth, hip, sil, db, q, N)
    ref_glsl = reference_glsl_fixture_check(capi, synth)
    if isinstance(ref_glsl, dict) and "error" not in ref_glsl:
        ref_glsl["baseline_sensor_size"] = reference_glsl_sample_check(capi, synth)
        ref_glsl["default_mode_bricks_on"] = reference_glsl_mode_check(capi, synth)
        ref_glsl["headline_grid_z_bands"] = reference_glsl_big_check(capi, synth, "four_sensors_512x424_into_512_bands")
        ref_glsl["default_mode_dxt1_bricks_at_sensor_size"] = reference_glsl_big_check(capi, synth, "default_mode_dxt1_bricks_512x424_into_128")
    what = "all %d z rows" % Z if done == Z else "%d of %d z rows, extrapolated to the grid" % (done, Z)
    return {"reference_cpu_work": ref_cpu, "reference_shader_text": ref_text, "reference_glsl_on_mesa": ref_glsl,
            "value": round(V_total / t_full / 1e6, 2), "unit": "Mvoxels/s", "cores": threads,
            "cpu_model": cpu_model(), "nproc": os.cpu_count(),      # the box's CPU and its logical CPU count
            "kind": "port",
            "sample": "oracle (OpenMP, %d threads = CPUs granted by affinity and cgroup quota) on the benchmark workload: full "
                      "pre_* chain of the %d-sensor frame (median of 3: %.2f s) + integrate of %s (%.2f s)"
                      % (threads, N, t_pre, what, t_int),
            "integrate_mvoxels_per_s": round(g.res_volume[0] * g.res_volume[1] * done / t_int / 1e6, 2),
            "parity_rows_bit_exact": parity, "parity_rows": done}


def reference_glsl_fixture_check(capi, synth, name="four_sensors_128x106_into_64"):
    """The HIP path against what the reference's OWN GLSL produced when Mesa llvmpipe ran it in the build container
    (tests/golden/gl_passes_<name>.npz: data, made by tests/golden/make_gl_golden.py; tolerances and caveats in
    tests/test_gl_ref.py / DESIGN.md section 2): the fixture's scene (4 sensors 128 x 106 into 64^3) through the
    library, largest absolute differences per output, and whether any voxel changes class.  Not timed, not the
    benchmark workload: it puts the parity against the reference's shaders into the bench record."""
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
        import shader_cases
        path = os.path.join(ROOT, "tests", "golden", "gl_passes_%s.npz" % name)
        if not os.path.exists(path):
            return None
        fx = np.load(path)
        scene, cfg, geo, inv, inv_res = shader_cases.build(synth, capi, name)
        if bytes(fx["inputs_sha256"]).decode() != shader_cases.digest(scene, inv):
            return {"error": "the synthetic scene drifted from the fixture's"}
        n = shader_cases.CASES[name][0]
        c = capi.Context(cfg, 0)
        for i in range(n):
            c.set_calibration(i, scene.xyz[i], scene.lut_res, scene.uv[i], scene.lut_res, (0.5, 4.5))
            c.set_inverse_calibration(i, inv[i], inv_res)
        c.set_use_bricks(False)
        c.step(scene.depth, scene.color)
        imgs = {"morph": 1, "depth_rg": 2, "lab": 3, "depth_b": 4, "sil": 5, "normal": 6, "quality": 7}
        out = {}
        for k, which in imgs.items():
            got = np.stack([c.readback_image(which, i) for i in range(n)]).astype(np.float64)
            want = fx[k].astype(np.float64)
            fin = np.isfinite(got) & np.isfinite(want)
            out[k] = float(np.abs(got - want)[fin].max())
        counters_equal = bool(np.array_equal(c.readback_brick_counters(), fx["counters"]))
        t, r = c.readback_tsdf(), fx["tsdf"]
        c.close()
        ok = ~(np.isnan(t) | np.isnan(r))
        lim = np.float32(cfg.tsdf_limit)

        def cls(v):
            return np.where(v <= -lim, -1, np.where(v >= lim, 1, 0))
        return {"what": "HIP path vs the reference's GLSL run on Mesa llvmpipe (committed fixture gl_passes_%s.npz)" % name,
                "max_abs_diff": {k: float("%.3g" % v) for k, v in out.items()}, "brick_counters_equal": counters_equal,
                "tsdf_max_abs_diff": float("%.3g" % np.abs(t.astype(np.float64) - r)[ok].max()),
                "tsdf_voxels_differing": int((t != r)[ok].sum()), "tsdf_voxels": int(t.size),
                "tsdf_voxels_changing_class": int((cls(t) != cls(r))[ok].sum()),
                "voxels_masked_nan_on_llvmpipe_only": int((np.isnan(r) & ~np.isnan(t)).sum()),
                "renderer": bytes(fx["gl_renderer"]).decode()}
    except Exception as e:  # noqa: BLE001 -- an extra key must never cost the line
        return {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}


def _tsdf_summary(t, r, limit):
    ok = ~(np.isnan(t) | np.isnan(r))
    lim = np.float32(limit)
    d = np.abs(t.astype(np.float64) - r)[ok]

    def cls(v):
        return np.where(v <= -lim, -1, np.where(v >= lim, 1, 0))
    return {"tsdf_max_abs_diff": float("%.3g" % (d.max() if d.size else 0.0)), "tsdf_voxels_beyond_5e-7": int((d > 5e-7).sum()),
            "tsdf_voxels_compared": int(ok.sum()), "tsdf_voxels_in_band": int((np.abs(r[ok]) < lim).sum()),
            "tsdf_voxels_changing_class": int((cls(t) != cls(r))[ok].sum())}


def reference_glsl_mode_check(capi, synth, name="bricks_reference_box_5_voxel_bricks"):
    """The library in the reference's DEFAULT mode (bricks on) against the Mesa run of the same mode: the reference's own
    box (-1,0,-1)-(1,2.2,1) with 5-voxel bricks that share rows, tsdf_integration.vs drawn through the occupied bricks'
    containedVoxels index lists (tests/golden/gl_passes_<name>.npz)."""
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
        import shader_cases
        path = os.path.join(ROOT, "tests", "golden", "gl_passes_%s.npz" % name)
        if not os.path.exists(path):
            return None
        fx = np.load(path)
        c = shader_cases.MODE_CASES[name]
        scene, cfg, geo, inv, inv_res = shader_cases.build_mode(synth, capi, name)
        if bytes(fx["inputs_sha256"]).decode() != shader_cases.digest_mode(scene, inv):
            return {"error": "the synthetic scene drifted from the fixture's"}
        ctx = capi.Context(cfg, 0)
        for i in range(c["n"]):
            ctx.set_calibration(i, scene.xyz[i], scene.lut_res, scene.uv[i], scene.lut_res, (0.5, 4.5))
            ctx.set_inverse_calibration(i, inv[i], inv_res)
        ctx.step(scene.depth, scene.color)
        out = {"what": "HIP path with RGBDR_FLAG_USE_BRICKS vs the reference's GLSL drawn through the occupied bricks' index lists on "
                       "Mesa (gl_passes_%s.npz: grid %s, %d of %d bricks occupied)" % (name, "x".join(str(v) for v in geo.res_volume),
                                                                                       fx["occupied"].size, fx["counters"].size),
               "brick_counters_equal": bool(np.array_equal(ctx.readback_brick_counters(), fx["counters"])),
               "occupied_bricks_equal": bool(np.array_equal(ctx.get_occupied()[0], fx["occupied"]))}
        out.update(_tsdf_summary(ctx.readback_tsdf(), fx["tsdf"], cfg.tsdf_limit))
        ctx.close()
        return out
    except Exception as e:  # noqa: BLE001 -- an extra key must never cost the line
        return {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}


def reference_glsl_big_check(capi, synth, name):
    """The larger Mesa samples of tests/golden/make_gl_golden.py BIG_SAMPLES: z bands of the 512^3 HEADLINE grid from four
    512 x 424 sensors; the default mode (DXT1 1280 x 1080 colour, bricks on) at that sensor size."""
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
        import make_gl_golden as mg
        path = os.path.join(ROOT, "tests", "golden", "gl_sample_%s.npz" % name)
        if not os.path.exists(path):
            return None
        fx = np.load(path)
        c = mg.BIG_SAMPLES[name]
        G = c["G"]
        scene, cfg, geo, inv = mg.big_scene(name)
        if bytes(fx["inputs_sha256"]).decode() != mg.big_digest(scene, inv, name):
            return {"error": "the synthetic scene drifted from the fixture's"}
        ctx = capi.Context(cfg, 0)
        for i in range(4):
            ctx.set_calibration(i, scene.xyz[i], scene.lut_res, scene.uv[i], scene.lut_res, (0.5, 4.5))
            a = inv[i]
            if a.shape[-1] == 3:
                full = np.zeros((G, G, G, 4), np.float32)
                for z0, z1 in c["bands"]:
                    full[z0:z1, ..., :3] = a[z0:z1]
                a = full
            ctx.set_inverse_calibration(i, a, (G, G, G))
            del a
        inv = None
        ctx.step(scene.depth, scene.color_blocks if c.get("dxt") else scene.color)
        tex = fx["texels"].astype(np.int64)
        n, H, W = 4, 424, 512
        imgs = {}
        for k, which in {"depth_rg": 2, "lab": 3, "depth_b": 4, "sil": 5, "normal": 6, "quality": 7}.items():
            got = np.stack([ctx.readback_image(which, i) for i in range(n)]).reshape(n * H * W, -1)[tex].astype(np.float64)
            want = fx[k].astype(np.float64)
            fin = np.isfinite(got) & np.isfinite(want)
            imgs[k] = float("%.3g" % np.abs(got - want)[fin].max())
        out = {"what": "%s: %d sampled texels per image (every edge-class texel of pre_boundary among them), %d sampled voxels" % (
                   name, tex.size, fx["voxels"].size),
               "max_abs_diff": imgs,
               "brick_counts_differing": int(np.abs(ctx.readback_brick_counters().astype(np.int64) - fx["counters"].astype(np.int64)).sum()),
               "brick_counts": int(fx["counters"].sum())}
        if "occupied" in fx.files:
            out["occupied_bricks_equal"] = bool(np.array_equal(ctx.get_occupied()[0], fx["occupied"]))
        t = ctx.readback_tsdf().reshape(-1)[fx["voxels"].astype(np.int64)]
        ctx.close()
        out.update(_tsdf_summary(t, fx["tsdf"], cfg.tsdf_limit))
        return out
    except Exception as e:  # noqa: BLE001 -- an extra key must never cost the line
        return {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}


def reference_glsl_sample_check(capi, synth, name="four_sensors_512x424_into_128"):
    """the same at BASELINE's sensor size: four 512 x 424 sensors into 128^3, against the committed SAMPLE of the Mesa run
    (tests/golden/gl_sample_<name>.npz: 19 814 texels of every image, 59 413 voxels, every brick counter)"""
    try:
        path = os.path.join(ROOT, "tests", "golden", "gl_sample_%s.npz" % name)
        if not os.path.exists(path):
            return None
        fx = np.load(path)
        G, n, H, W = 128, 4, 424, 512
        scene = synth.Scene(n, W, H, lut_res=(32, 27, 32), seed=1234)
        cfg = capi.make_config(n, (W, H), voxel_size=2.0 / G, brick_size=8 * 2.0 / G)
        inv = scene.inverse((G, G, G))
        c = capi.Context(cfg, 0)
        for i in range(n):
            c.set_calibration(i, scene.xyz[i], scene.lut_res, scene.uv[i], scene.lut_res, (0.5, 4.5))
            c.set_inverse_calibration(i, inv[i], (G, G, G))
        c.set_use_bricks(False)
        c.step(scene.depth, scene.color)
        tex = fx["texels"].astype(np.int64)
        out = {}
        for k, which in {"morph": 1, "depth_rg": 2, "lab": 3, "depth_b": 4, "sil": 5, "normal": 6, "quality": 7}.items():
            got = np.stack([c.readback_image(which, i) for i in range(n)]).reshape(n * H * W, -1)[tex].astype(np.float64)
            want = fx[k].astype(np.float64)
            fin = np.isfinite(got) & np.isfinite(want)
            out[k] = float("%.3g" % np.abs(got - want)[fin].max())
        cnt = c.readback_brick_counters().astype(np.int64)
        t = c.readback_tsdf().reshape(-1)[fx["voxels"].astype(np.int64)]
        c.close()
        r = fx["tsdf"]
        ok = ~(np.isnan(t) | np.isnan(r))
        lim = np.float32(cfg.tsdf_limit)
        return {"what": "4 sensors 512 x 424 into 128^3, %d sampled texels per image, %d sampled voxels (%d in the band)"
                        % (tex.size, t.size, int((np.abs(r[ok]) < lim).sum())),
                "max_abs_diff": out, "brick_counts_differing": int(np.abs(cnt - fx["counters"].astype(np.int64)).sum()),
                "brick_counts": int(fx["counters"].sum()),
                "tsdf_max_abs_diff": float("%.3g" % np.abs(t.astype(np.float64) - r)[ok].max()),
                "tsdf_voxels_beyond_1e-6": int((np.abs(t.astype(np.float64) - r)[ok] > 1e-6).sum())}
    except Exception as e:  # noqa: BLE001 -- an extra key must never cost the line
        return {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}


def reference_text_baseline(ctx, scene, capi, synth, hip, sil, db, q, N, rows=32):
    """The TEXT of the reference's own shaders compiled as C++ (oracle/_ref/libref_shaders.so, built in the build
    container by oracle/build_shader_ref.py; samplers are stand-ins, see oracle/glsl_runtime.hpp), one thread, on a
    bounded sample of the benchmark workload: the pre_* chain of sensor 0 and tsdf_integration.vs on `rows` z rows in
    the middle of the volume -- timed, and compared bit for bit with the HIP images / volume rows.  None where the
    library did not travel (it exists only where /root/reference was present at build time)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    try:
        import shader_ref
        if not shader_ref.available():
            return None
        g = ctx.geo
        X, Y, Z = g.res_volume
        z0 = (Z // 2 // 8) * 8
        inv = [ctx.readback_inverse_calibration(i, z0, z0 + rows) for i in range(N)]
        # tsdf_integration.vs over the rows: a 1:1 LUT is looked up at texel centres, so the rows are a volume of their own
        tsdf = np.full((rows, Y, X), -np.float32(0.01), np.float32)
        itg = shader_ref.Shader("tsdf_integration")
        for i in range(N):
            itg.volume("cv_xyz_inv", inv[i], i)
        itg.array_f32("kinect_silhouettes", np.stack(sil), linear=True)
        itg.array_f32("kinect_depths", np.stack(db), linear=False)
        itg.array_f32("kinect_qualities", np.stack(q), linear=True)
        itg.f("limit", 0.01)
        itg.u("num_kinects", N)
        itg.set("res_tsdf", np.array([X, Y, rows], np.uint32))
        itg.set("volume_tsdf", shader_ref.Image3D(tsdf.ctypes.data, X, Y, rows))
        t0 = time.perf_counter()
        itg.run(X, Y, rows, 0, rows)
        t_int = time.perf_counter() - t0
        got = hip[z0:z0 + rows]
        same_vol = bool(np.all((tsdf == got) | (np.isnan(tsdf) & np.isnan(got))))

        class One:                                     # sensor 0 alone through the pre_* shader text
            pass

        one = One()
        one.N, one.depth, one.color, one.xyz, one.uv = 1, scene.depth[:1], scene.color[:1], scene.xyz[:1], scene.uv[:1]
        t0 = time.perf_counter()
        frame = shader_ref.run_frame(one, synth.BBOX_MIN, synth.BBOX_MAX, (X, Y, Z), None, brick_size=g.brick_size,
                                     res_bricks=tuple(g.res_bricks))
        t_pre = time.perf_counter() - t0
        same_img = all(bool(np.all((frame[k][0] == ctx.readback_image(w, 0)) | (np.isnan(frame[k][0]) & np.isnan(ctx.readback_image(w, 0)))))
                       for k, w in (("depth_b", capi.IMG_DEPTH_B_RG), ("sil", capi.IMG_SILHOUETTE), ("quality", capi.IMG_QUALITY),
                                    ("normal", capi.IMG_NORMAL), ("lab", capi.IMG_LAB)))
        return {"what": "the reference's shader text compiled as C++ (stand-in samplers), 1 thread: tsdf_integration.vs on %d of %d "
                        "z rows, pre_* chain of 1 of %d sensors" % (rows, Z, N),
                "integrate_mvoxels_per_s": round(X * Y * rows / t_int / 1e6, 2), "integrate_s": round(t_int, 2),
                "pre_chain_one_sensor_s": round(t_pre, 2),
                "hip_rows_bit_identical": same_vol, "hip_images_bit_identical": same_img, "rows": rows}
    except Exception as e:  # noqa: BLE001 -- an extra key must never cost the line
        return {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}


if __name__ == "__main__":
    main()
