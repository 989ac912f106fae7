"""bench.py's choice of the pre_* chain schedule of an N > 1 run: sharded by sensor, redundant, or sharded and one
frame ahead
of the sweep (dist.LaggedChain) -- timed on the run's own ranks before the headline."""
import os
import sys
import time


def all_ranks_ok(rig, ok):
    """MIN over the ranks of a 0 / 1 flag, through the gloo side group (every rank takes the same road)"""
    if rig.world == 1:
        return bool(ok)
    flag = rig.torch.tensor([1 if ok else 0], dtype=rig.torch.int32)
    # (the gloo side group next to an nccl job; a gloo job's own default group carries host tensors)
    rig.dist.all_reduce(flag, op=rig.dist.ReduceOp.MIN, group=rig.shared.get("fallback"))
    return bool(int(flag[0]))


def choose_chain(rig, steps=40, warmup=8):
    """Which schedule for the pre_* chain of an N > 1 run?
      sharded    rank r runs n / k sensors; one all-gather of the packed frames + one all-reduce of the brick counters
      sit
                 between chain and sweep (3.5 MB per rank at configs[3]: ~20 us to the same GPU, an estimated 45-140 us
                 over
                 xGMI, against 60-70 us of chain time saved);
      redundant  every rank runs every sensor, no collective;
      lagged     sharded on a chain-only context one frame AHEAD of the sweep, so the gather of frame k+1 travels under
                 the sweep of frame k (dist.LaggedChain, rgbdr_import_frame): one chain + one gather + one sweep per
                 step,
                 like the others, one frame of latency more.
    Which is shortest depends on the interconnect, so the run MEASURES all three on its own ranks before the headline
    and keeps the fastest (max over ranks); the line records the three times.  (Timed like the headline -- host clock
    between barriers -- and over as many steps as a short headline: at 12 steps the ~1 ms it takes to fill and drain the
    queues weighed 0.1 ms per step and differently per schedule; the trials then preferred the lagged one, 0.741 against
    0.761 ms, where 40 pinned steps gave 0.665 against 0.637: profiles/hwq_ab.sh.)"""
    if rig.gather is None or not rig.multi:
        return
    torch, dist, ctx, capi, rdist = rig.torch, rig.dist, rig.ctx, rig.capi, rig.rdist
    n = ctx.cfg.num_sensors
    keep_gather, first, count = rig.gather, n // rig.slab_count * rig.slab_rank, n // rig.slab_count
    ctx.set_use_bricks(False)                # the headline's sweep

    def run():
        # (a trial that contains the collective library's one-off host stall is run again)
        for attempt in range(3):
            for _ in range(warmup):
                rig.step(False)
            rig.barrier()
            t0 = time.perf_counter()
            t_prev, longest = t0, 0.0
            for _ in range(steps):
                rig.step(False)
                t_now = time.perf_counter()
                longest = max(longest, t_now - t_prev)
                t_prev = t_now
            rig.barrier()
            t = torch.tensor([(time.perf_counter() - t0) / steps * 1e3], dtype=torch.float64)
            if rig.world > 1:
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=rig.shared.get("fallback"))
            if all_ranks_ok(rig, longest * 1e3 <= STALL_MS):
                break
        return float(t[0])

    # (host-staged debugging transport: tens of ms per step -- short trials)
    slow_transport = rig.args.backend == "gloo"
    if slow_transport:
        steps, warmup = 6, 2
    # clocks up first: the schedule tried first is not to pay the ramp (3 % over ~25 frames)
    for _ in range(0 if slow_transport else 60):
        rig.step(False)
    times = {"sharded": run()}
    ctx.set_sensor_shard(0, 0)
    rig.gather = None
    times["redundant"] = run()
    # lagged: a chain-only context (same sensors, box and brick size -> the same brick grid; one voxel per brick)
    lag = chain = None
    try:
        if not getattr(rig.args, "lagged", True):
            raise RuntimeError("--no-lagged")
        g = rig.geo
        chain = capi.Context(capi.make_config(n, (rig.W, rig.H), voxel_size=g.brick_size, brick_size=g.brick_size),
            rig.local_rank)
        if tuple(chain.geo.res_bricks) != tuple(g.res_bricks) or chain.geo.brick_size != g.brick_size:
            raise RuntimeError("the chain-only context's brick grid differs")
        for i in range(n):
            chain.set_calibration(i, rig.scene.xyz[i], rig.scene.lut_res, rig.scene.uv[i], rig.scene.lut_res, (0.5,
                4.5))
        # (dist.FrameGather loopback: the brick counts of an unsharded frame stand for the other ranks')
        if rig.loop:
            chain.update_device(rig.d_depth.data_ptr(), rig.d_color.data_ptr())
            chain.clear_occupied_bricks(); chain.process_textures()
            chain.sync()
        raw = getattr(rig.exchanger, "comm_gather", None) if rig.managed and not rig.loop else None
        # the library enqueues the gather itself (rgbdr_shard_allgather_async) on the raw communicator
        if raw is not None:
            chain.set_sensor_shard(first, count)
            lag_gather = None
        elif rig.managed and getattr(rig.exchanger, "comm_gather", None) is not None:     # one GPU, raw RCCL to itself
            lag_gather = rdist.RawLoopbackGather(chain, rig.dev, rig.slab_rank, rig.slab_count,
                rig.exchanger.comm_gather)
        # torch.distributed's collectives on a side stream, or the one-GPU loopback
        else:
            lag_gather = rdist.FrameGather(chain, rig.dev, rank=rig.slab_rank, world=rig.slab_count,
                group=rig.transport["group"],
                                           via_host=rig.transport["kind"] != "rccl", loopback=rig.loop)
        lag = rdist.LaggedChain(ctx, chain, rig.dev, lag_gather,
                                before_sweep=rig.exchanger.begin_step if rig.halo is not None else None,
                                after_sweep=rig.exchanger.exchange_async if rig.halo is not None else None,
                                nccl_comm=raw.handle if raw is not None else None,
                                sweep_launches=int(os.environ.get("RGBDR_BENCH_LAG_LAUNCHES", "2")))
        rig.lag = lag
        times["lagged"] = run()
        lag.flush()
        rig.barrier()
    except Exception as e:  # noqa: BLE001 -- a schedule that does not come up is not a candidate
        sys.stderr.write("[bench rank %d] lagged chain unavailable (%s: %s)\n" % (rig.rank, type(e).__name__,
            str(e)[:200]))
        times["lagged"] = None
    rig.lag = None
    ctx.set_sweep_launches(1)
    ok = all_ranks_ok(rig, times["lagged"] is not None)
    cands = {k: v for k, v in times.items() if v is not None and (k != "lagged" or ok)}
    kept = min((k for k in cands if k != "lagged"), key=cands.get)
    # one frame of latency more: only for a gain beyond the noise
    if "lagged" in cands and cands["lagged"] < 0.98 * cands[kept]:
        kept = "lagged"
    if os.environ.get("RGBDR_BENCH_CHAIN") in cands:        # pin the choice (tests, A/B runs)
        kept = os.environ["RGBDR_BENCH_CHAIN"]
    if kept == "sharded":
        ctx.set_sensor_shard(first, count)
        rig.gather = keep_gather
    elif kept == "lagged":
        rig.lag = lag
        ctx.set_sweep_launches(lag.sweep_launches)
    if kept != "lagged" and chain is not None:
        chain.close()
        lag = None
    # the legs after the headline run on the better of the two plain schedules
    rig.plain_chain = ("sharded", keep_gather, first,
        count) if times["sharded"] <= times["redundant"] else ("redundant", None, 0, 0)
    rig.lag_keep = (lag, chain)
    rig.chain_choice = {"ms_per_step_sharded": round(times["sharded"], 4),
        "ms_per_step_redundant": round(times["redundant"], 4),
                        "ms_per_step_lagged": round(times["lagged"], 4) if times["lagged"] is not None else None,
                        "kept": kept, "steps_each": steps}


def leave_lagged_chain(rig):
    """after the headline: sweep the frame still pending, close the chain-only context and put the legs on the better of
    the two plain schedules"""
    lag, chain = getattr(rig, "lag_keep", (None, None))
    if rig.lag is None:
        return
    rig.lag.close()
    rig.barrier()
    rig.lag = None
    kind, gather, first, count = rig.plain_chain
    if kind == "sharded":
        rig.ctx.set_sensor_shard(first, count)
        rig.gather = gather
    if chain is not None:
        chain.close()


def recheck_lagged_headline(rig, dt, stats):
    """The lagged schedule was kept on its trial, but its K timed steps took 10 % longer per step than the trial AND
    longer
    than the better plain schedule's trial (seen on one GPU: a single 50-100 ms stall of the host inside RCCL's enqueue
    under this schedule, profiles/r05_notes/scaling_tail.md): leave it, time the K steps again on the plain schedule and
    report those -- the line keeps what the discarded attempt read.  (dt and the trials are maxima over the ranks: every
    rank decides alike.)"""
    c = rig.chain_choice
    ms = dt / rig.args.steps * 1e3
    plain = min(c["ms_per_step_sharded"], c["ms_per_step_redundant"])
    if c.get("ms_per_step_lagged") is None or (ms <= 1.10 * c["ms_per_step_lagged"] or ms <= plain):
        return dt, stats
    leave_lagged_chain(rig)
    c["lagged_headline_discarded_ms_per_step"] = round(ms, 4)
    c["kept"] = rig.plain_chain[0]
    c["why"] = ("the lagged schedule's timed steps took more than 1.1 x its trial: "
                "timed again on the better plain schedule")
    dt, stats = rig.timed(False, rig.args.steps, rig.args.warmup)
    rig.stats = stats
    return dt, stats


STALL_MS = 30.0


def retime_after_a_host_stall(rig, dt, stats):
    """Once per process (70-300 frames in) the host is held for 36-100 ms inside RCCL's enqueue -- no HIP call in
    progress, whichever schedule runs, the device idle meanwhile (profiles/r05_notes/scaling_tail.md,
    lag_stall_probe.py,
    stall_log.sh).  A run of K = 40 steps that contains it reads two to three times its steady state.  The criterion is
    the host's own clock: a single step of the timed loop that took the host more than 30 ms (a frame is 0.6-2.4 ms) on
    ANY rank -> every rank times its K steps again, at most twice; the line keeps what the discarded attempts read."""
    discarded = []
    for _ in range(2):
        stalled = getattr(rig, "longest_host_step_ms", 0.0) > STALL_MS
        if all_ranks_ok(rig, not stalled):
            break
        discarded.append({"ms_per_step": round(dt / rig.args.steps * 1e3, 4),
            "longest_host_step_ms": round(rig.longest_host_step_ms, 1)})
        dt, stats = rig.timed(False, rig.args.steps, rig.args.warmup)
        rig.stats = stats
    if discarded:
        rig.retimed = {"why": "a step of the timed loop held the host for more than %.0f ms on some rank (a one-off "
            "stall "
                              "inside the collective library's enqueue): the K steps were timed again" % STALL_MS,
                       "discarded": discarded, "longest_host_step_ms_kept": round(rig.longest_host_step_ms, 1)}
    return dt, stats

