"""bench_legs.py -- every extra key of bench.py's JSON line, measured AFTER the headline.

A leg is a function of the Rig (bench.py) that returns the value of its key.  run_all() runs each one under the
watchdog and inside try/except: a leg that throws leaves {"error": ...} under its key, a leg that hangs is ended by the
watchdog, which prints the line as far as it got -- the headline (value, roofline) never depends on a leg.

Data dependence: the headline is a full sweep with no cross-frame state, but the brick-skipping
mode, the background skip, store elision and the pre_* chain depend on what the sensors see and on what changed since
the last frame.  Those are therefore reported on three scenes: the static ring scene of SURVEY 8(d) (best case), a
DENSE scene (every pixel valid and inside the box) and a MOVING sequence (four different frames in rotation).
"""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))


def run_leg(rig, out, key, fn, budget=120.0, into=None):
    """one isolated leg: out[key] = fn() or {"error": ...}; RGBDR_BENCH_FAIL_LEG=<key> / <key>:hang are test hooks"""
    hook = os.environ.get("RGBDR_BENCH_FAIL_LEG", "")
    target = out if into is None else into
    if os.environ.get("RGBDR_BENCH_LEG_BUDGET"):          # tests: every leg gets this many seconds
        budget = float(os.environ["RGBDR_BENCH_LEG_BUDGET"])
    try:
        with rig.watchdog.phase("leg " + key, budget):
            if hook == key:
                raise RuntimeError("RGBDR_BENCH_FAIL_LEG=" + key)
            if hook == key + ":hang":
                time.sleep(1e6)
            target[key] = fn()
    except Exception as e:  # noqa: BLE001 -- an extra key must never cost the line
        target[key] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        sys.stderr.write("[bench rank %d] leg '%s' failed: %s: %s\n" % (rig.rank, key, type(e).__name__, str(e)[:300]))
        rig.restore_defaults()


def run_all(rig, out, lean=False):
    single = rig.world == 1 and not rig.loop
    if rig.loop:
        run_leg(rig, out, "slab", lambda: leg_slab(rig, out))
    if lean:
        out["bricked"] = None
        if rig.multi:
            run_leg(rig, out, "halo", lambda: leg_halo(rig, out))
        return
    run_leg(rig, out, "passes_ms", lambda: leg_breakdown(rig, out))
    run_leg(rig, out, "bricked", lambda: leg_bricked(rig))
    for key in ("other_schedule", "full_sweep_store_elision", "full_sweep_background_skip"):
        out[key] = None
    if single:
        run_leg(rig, out, "other_schedule", lambda: leg_other_schedule(rig))
        run_leg(rig, out, "full_sweep_store_elision", lambda: leg_elision(rig))
        run_leg(rig, out, "full_sweep_background_skip", lambda: leg_background_skip(rig))
        run_leg(rig, out, "box", lambda: leg_box(rig), into=out["roofline"])
        run_leg(rig, out, "scenes", lambda: leg_scenes(rig), budget=240.0)
        run_leg(rig, out, "post_pass", lambda: leg_post_pass(rig))
        run_leg(rig, out, "host_fed", lambda: leg_host_fed(rig))
        run_leg(rig, out, "reference_defaults", lambda: leg_reference_defaults(rig))
        run_leg(rig, out, "default_display_frame", lambda: leg_default_display_frame(rig))
        run_leg(rig, out, "inverse_lut", lambda: leg_inverse_lut(rig))
        if rig.rank == 0 and not rig.args.no_cpu_baseline:
            run_leg(rig, out, "cpu_baseline", lambda: cpu_baseline(rig), budget=300.0)
    if rig.multi:
        run_leg(rig, out, "halo", lambda: leg_halo(rig, out))
        run_leg(rig, out, "halo_copy_engine", lambda: leg_halo_copy_engine(rig, out))
        run_leg(rig, out, "post_pass", lambda: leg_post_pass_slabs(rig))


# ---- legs on the headline context -----------------------------------------------------------------------------------
def leg_breakdown(rig, out):
    """per-pass times, not part of the headline timing: the totals from a run with the three total timers, the five
    passes from a run with every timer (their event records inflate the totals)"""
    stats = dict(rig.stats)
    _, tot_stats = rig.timed(False, 5, 1, detail=1)
    _, pass_stats = rig.timed(False, 5, 1, detail=2)
    stats.update({k: v for k, v in tot_stats.items() if k not in stats})
    stats.update({k: v for k, v in pass_stats.items() if k not in stats})
    return {k: round(v[0] / max(v[1], 1) * 1e-6, 4) for k, v in stats.items()}


def leg_bricked(rig):
    """brick-skipping mode (the reference's default): integrate visits the occupied bricks only"""
    ctx, args = rig.ctx, rig.args
    bsteps = max(args.steps // 2, 1)
    dtb, stats_b = rig.timed(True, bsteps, 2)
    bint_ns, bint_n = stats_b["2integrate"]
    res = {"ms_per_step": round(dtb / bsteps * 1e3, 4), "value": round(rig.V_total / (dtb / bsteps) / 1e6, 1),
           "integrate_ms": round(bint_ns / max(bint_n, 1) * 1e-6, 4), "occupied_ratio": round(ctx.occupied_ratio(), 4),
           "ms_per_step_pipelined": None, "scene": "static ring scene (best case; dense / moving under `scenes`)"}
    # ... with the pre_* chain of frame k+1 overlapping the sweep of frame k (RGBDR_FLAG_PIPELINE): the sweep is short
    # here, so the two streams overlap for most of it
    if rig.world == 1 and not rig.loop and not args.pipeline:
        ctx.set_pipelined(True)
        try:
            dtbp, _ = rig.timed(True, bsteps, 2)
        finally:
            ctx.set_pipelined(False)
        res["ms_per_step_pipelined"] = round(dtbp / bsteps * 1e3, 4)
    ctx.set_use_bricks(False)
    return res


def leg_other_schedule(rig):
    """whichever of sequential / pipelined the headline did not use"""
    ctx, args = rig.ctx, rig.args
    ctx.set_pipelined(not args.pipeline)
    try:
        dto, stats_o = rig.timed(False, args.steps, args.warmup)
    finally:
        ctx.set_pipelined(bool(args.pipeline))
    oi_ns, oi_n = stats_o["2integrate"]
    return {"schedule": "sequential" if args.pipeline else "pipelined (pre_* of step k+1 on a second stream under "
        "integrate of step k)",
            "ms_per_step": round(dto / args.steps * 1e3, 4), "value": round(rig.V_total / (dto / args.steps) / 1e6, 1),
            "integrate_ms": round(oi_ns / max(oi_n, 1) * 1e-6, 4)}


def leg_elision(rig):
    """RGBDR_FLAG_ELIDE_STORES: the full sweep without re-storing tiles that stay -limit"""
    ctx, args = rig.ctx, rig.args
    ctx.set_elide_stores(True)
    try:
        dte, stats_e = rig.timed(False, args.steps, args.warmup)
    finally:
        ctx.set_elide_stores(False)
    ei_ns, ei_n = stats_e["2integrate"]
    return {"ms_per_step": round(dte / args.steps * 1e3, 4), "value": round(rig.V_total / (dte / args.steps) / 1e6, 1),
            "integrate_ms": round(ei_ns / max(ei_n, 1) * 1e-6, 4),
            "scene": "static ring scene (best case; dense / moving under `scenes`)"}


def skip_summary(rig, ctx, dts, stats_s, steps):
    skipped, total = ctx.skipped_pairs()
    verdicts = ctx.readback_skip_tables(0)
    si_ns, si_n = stats_s["2integrate"]
    listed = int((verdicts == 0).any(axis=1).sum())
    # bytes a steady-state sweep asks for: per pair the four words the classifier reads; per listed tile its
    # list entry, its TSDF store and the LUT planes of its undecided sensors; the frame texels (windows) once
    nbytes = int(total * 16 + listed * (8 + 2048) + (total - skipped) * 3 * 512 * 4 + rig.N * rig.W * rig.H * 8)
    return {"ms_per_step": round(dts / steps * 1e3, 4), "value": round(rig.V_total / (dts / steps) / 1e6, 1),
            "integrate_ms": round(si_ns / max(si_n, 1) * 1e-6, 4),
            "pairs_decided": int(skipped), "pairs": int(total), "frac_decided": round(skipped / max(total, 1), 4),
            "verdicts": {k: int((verdicts == i).sum()) for i, k in enumerate(("none", "carve", "in_front", "hidden"))},
            "tiles_listed": listed, "tiles": int(verdicts.shape[0]),
            "bytes_per_launch": nbytes, "GBps": round(nbytes / (si_ns / max(si_n, 1)), 1)}


def leg_background_skip(rig):
    """RGBDR_FLAG_SKIP_BACKGROUND: LUT planes of (tile, sensor) pairs whose frame window decides the outcome stay
    unread,
    tiles that are constants are not rewritten while they hold their constant"""
    ctx, args = rig.ctx, rig.args
    ctx.set_skip_background(True)
    try:
        dts, stats_s = rig.timed(False, args.steps, args.warmup)
        res = skip_summary(rig, ctx, dts, stats_s, args.steps)
    finally:
        ctx.set_skip_background(False)
    res["scene"] = "static ring scene (best case; dense / moving under `scenes`)"
    return res


def leg_slab(rig, out):
    """--slab / --loopback: what the staging costs the sweep -- the same slab without a staging set (plain kernel, no
    exchange) -- and this slab's row of a --slab-sweep table"""
    ctx, args, g = rig.ctx, rig.args, rig.geo
    halo_keep, rig.halo = rig.halo, None
    try:
        ctx.set_halo_staging(-1)
        dt_plain, stats_plain = rig.timed(False, args.steps, args.warmup)
    finally:
        rig.halo = halo_keep
    plain_ms = (stats_plain["2integrate"][0] / max(stats_plain["2integrate"][1], 1) * 1e-6, dt_plain / args.steps * 1e3)
    gather = rig.gather
    return {"rank": rig.slab_rank, "of": rig.slab_count, "owned_z_rows": int(g.slab_voxel_z1 - g.slab_voxel_z0),
            "faces_staged": int(rig.slab_rank > 0) + int(rig.slab_rank < rig.slab_count - 1),
            "integrate_ms": round(rig.int_s * 1e3, 4), "integrate_ms_without_staging": round(plain_ms[0], 4),
            "staging_overhead_ms": round(rig.int_s * 1e3 - plain_ms[0], 4),
            "ms_per_step": round(rig.ms_per_step, 4), "ms_per_step_without_halo": round(plain_ms[1], 4),
            "host_enqueue_ms_per_step": out["host_enqueue_ms_per_step"],
            "roofline_frac": round(rig.achieved / rig.HBM_PEAK, 4), "halo_ms_to_self": rig.halo_ms,
            "frame_gather_ms_to_self": gather.last_ms() if hasattr(gather, "last_ms") else None,
            "schedule": ("pipelined" if args.pipeline else "sequential") + (
                ", " + rig.chain_choice["kept"] + " chain" if getattr(rig, "chain_choice",
                None) else (", sharded chain" if gather is not None else "")) +
                        (", library-managed RCCL" if rig.managed else "") + (
                            ", RGBDR_CU_SPLIT="
                            + os.environ["RGBDR_CU_SPLIT"] if os.environ.get("RGBDR_CU_SPLIT") else "")}


def leg_halo(rig, out):
    rig.torch.cuda.synchronize()
    per_rank = out.get("per_rank")
    return {"layers_per_face": int(rig.geo.halo_tile_layers), "bytes_per_face": int(rig.halo[0].numel() * 4),
            "transfer_ms_rank0": rig.exchanger.last_transfer_ms(),
            "transfer_ms_max": max([h for h in per_rank["halo_ms"] if h is not None],
            default=None) if per_rank else rig.halo_ms}


def leg_halo_copy_engine(rig, out):
    """The same steps with the halo moved by the COPY ENGINE instead of RCCL's send / recv kernels (the C ABI's
    rgbdr_halo_export / _set_peer / _pull_async; dist.PeerCopySlabExchange): every rank pulls its neighbours' staged
    faces
    with device-to-device copies from their IPC-mapped staging sets.  RCCL stays the transport of the headline; this leg
    says what the alternative gives on the same ranks (the gather of a sharded chain still runs over RCCL)."""
    ctx, args = rig.ctx, rig.args
    keep = rig.exchanger
    keep.wait()
    rig.barrier()
    # exports are Python bytes: a gloo group carries them
    group = rig.shared.get("fallback") if rig.world > 1 else None
    peer = rig.rdist.PeerCopySlabExchange(ctx, rig.dev, rig.slab_rank, rig.slab_count, group=group, loopback=rig.loop)

    class Both:                      # halo by copy engine, everything else (the gather) as before
        begin_step, exchange_async = peer.begin_step, peer.exchange_async
        wait, last_transfer_ms = peer.wait, peer.last_transfer_ms
    rig.exchanger = Both()
    try:
        steps = max(8, min(args.steps, 40))
        # (every rank the same number of steps: a loop by the clock would leave the neighbours of the rank that stops
        # first
        # waiting for faces that never come)
        dt, st = rig.timed(False, steps, 24 if rig.args.backend == "nccl" else 4)
        ctx.enable_timers(True)
        ctx.set_timer_detail(2)
        rig.step(False)
        rig.step(False)
        peer.wait()
        ctx.sync()
        ms = peer.last_transfer_ms()
        ctx.enable_timers(False)
        # what the headline ran on
        head = "copy_engine" if getattr(rig, "halo_by",
            "rccl") == "peer" else ("rccl" if rig.transport["kind"] == "rccl" else "host_staged")
        return {"halo_transport": "copy engine: hipMemcpyAsync from the neighbours' IPC-mapped staging sets behind "
            "step words the streams "
                                  "write / wait for (hipStreamWriteValue32 / hipStreamWaitValue32)"
                                  + (" (loopback: this process is its own neighbour, no IPC)" if rig.loop else ""),
                "ms_per_step": round(dt / steps * 1e3, 4), "ms_per_step_headline_" + head: out["ms_per_step"],
                "integrate_ms": round(st["2integrate"][0] / max(st["2integrate"][1], 1) * 1e-6, 4),
                "integrate_ms_headline_" + head: round(rig.int_s * 1e3, 4),
                "transfer_ms": None if ms is None else round(ms, 4), "transfer_ms_headline_" + head: rig.halo_ms,
                "host_enqueue_ms_per_step": round(rig.host_enqueue_ms, 4),
                "longest_host_step_ms": round(rig.longest_host_step_ms, 3),
                "sweep_launches": 1}
    finally:
        rig.exchanger = keep
        try:
            peer.close()
        except Exception:  # noqa: BLE001
            pass
        rig.barrier()


# ---- clocks and power under load -----------------------------------------------------------------------------------
def gpu_sysfs(torch, index=0):
    """the sysfs directory of HIP device `index`: matched by PCI address (a box shows the cards of every GPU of the
    node,
    the process sees one of them); the first card with clocks when the address cannot be matched"""
    import glob
    cards = [d for d in sorted(glob.glob("/sys/class/drm/card*/device")) if os.path.exists(d + "/pp_dpm_mclk")]
    try:
        p = torch.cuda.get_device_properties(index)
        bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        for d in cards:
            if os.path.basename(os.path.realpath(d)) == bdf:
                return d, bdf
    except Exception:  # noqa: BLE001
        pass
    return (cards[min(index, len(cards) - 1)], None) if cards else (None, None)


def read_box(dev):
    """one sample of clocks / power / busy of a GPU from sysfs (no child process: under `rocprofv3 --pmc` every child
    inherits the profiler's preloaded library, and rocm-smi is a script that re-executes itself after it)"""
    import glob
    s = {}
    for name in ("sclk", "mclk"):
        try:
            cur = [ln for ln in open("%s/pp_dpm_%s" % (dev, name)).read().splitlines() if ln.rstrip().endswith("*")]
            if cur:
                s[name + "_MHz"] = float(cur[0].split(":", 1)[1].strip(" *").lower().replace("mhz", ""))
        except (OSError, ValueError):
            pass
    for f in glob.glob(dev + "/hwmon/hwmon*/freq1_input"):
        try:
            s["sclk_hwmon_MHz"] = int(open(f).read()) / 1e6
            break
        except (OSError, ValueError):
            pass
    for pw in glob.glob(dev + "/hwmon/hwmon*/power1_average") + glob.glob(dev + "/hwmon/hwmon*/power1_input"):
        try:
            s["power_W"] = int(open(pw).read()) / 1e6
            break
        except (OSError, ValueError):
            pass
    try:
        s["gpu_busy_percent"] = float(open(dev + "/gpu_busy_percent").read())
    except (OSError, ValueError):
        pass
    return s


class BoxSampler(threading.Thread):
    def __init__(self, dev, period=0.02):
        super().__init__(daemon=True)
        self.dev, self.period, self.samples, self.stop_flag = dev, period, [], False

    def run(self):
        while not self.stop_flag:
            self.samples.append((time.perf_counter(), read_box(self.dev)))
            time.sleep(self.period)


def leg_box(rig):
    """roofline.box: clocks, power and busy percentage of the GPU UNDER LOAD -- an untimed burst of the same step loop
    of
    at least one second with sysfs sampled from a side thread every 20 ms; the statistics are over the samples of the
    middle 60 % of the burst.  (Round 4 took one sample after enqueueing a 23 ms timed region and read an idle clock.) 
    A
    clock that still reads idle next to a busy GPU is a stale sysfs marker, and is dropped rather than printed."""
    dev, bdf = gpu_sysfs(rig.torch, rig.local_rank)
    if dev is None:
        return {"unavailable": "no GPU with pp_dpm_mclk under /sys/class/drm"}
    try:
        level = open(dev + "/power_dpm_force_performance_level").read().strip()
    except OSError:
        level = None
    steps = int(1.2 / (rig.ms_per_step * 1e-3)) + 1          # at least a second of the headline's step loop
    if profiled():
        return {"unavailable": "a profiler is preloaded: the one-second burst would fill its trace"}
    sampler = BoxSampler(dev)
    rig.ctx.sync()
    sampler.start()
    t0 = time.perf_counter()
    marks = []
    for i in range(steps):
        rig.step(False)
        if i % 64 == 63:
            # keep the queue short: the burst then lasts as long on the host as on the GPU
            rig.ctx.sync()
            marks.append((i + 1, time.perf_counter()))
    rig.ctx.sync()
    t1 = time.perf_counter()
    sampler.stop_flag = True
    sampler.join(1.0)
    lo, hi = t0 + 0.2 * (t1 - t0), t0 + 0.8 * (t1 - t0)
    mid = [s for t, s in sampler.samples if lo <= t <= hi]

    def stat(key):
        v = sorted(s[key] for s in mid if key in s)
        return {"min": round(v[0], 1), "median": round(v[len(v) // 2], 1), "max": round(v[-1], 1)} if v else None

    # The headline's 20-50 steps start from an idle GPU; under sustained load the package reaches its power limit and
    # the
    # clock settles lower.  The burst's second half says what a long-running loop gets per step.
    half = [m for m in marks if m[0] >= steps // 2]
    sustained = (half[-1][1] - half[0][1]) / (half[-1][0] - half[0][0]) * 1e3 if len(half) >= 2 else None
    res = {"burst_s": round(t1 - t0, 3), "burst_steps": steps, "samples": len(mid), "perf_level": level,
           "sustained_ms_per_step": round(sustained, 4) if sustained else None,
           "sustained_vs_headline": round(sustained / rig.ms_per_step, 4) if sustained else None,
           "sysfs": dev, "pci": bdf if bdf else "not matched: the first card with clocks",
           "power_W": stat("power_W"), "mclk_MHz": stat("mclk_MHz")}
    busy = stat("gpu_busy_percent")
    if busy and busy["max"] > 0:
        res["gpu_busy_percent"] = busy
    else:      # 0 throughout a burst means the card read is not the one that ran it: not evidence of anything
        res["gpu_busy_percent"] = None
    sclk = stat("sclk_hwmon_MHz") or stat("sclk_MHz")
    if sclk and sclk["max"] >= 500.0:
        res["sclk_MHz"] = sclk
    else:
        res["sclk_MHz"] = None
        res["sclk_note"] = ("dropped: sysfs reads %s MHz during the burst (a stale DPM marker, not the clock the "
                            "kernels ran at)" % (sclk["max"] if sclk else "nothing"))
    return res


# ---- the data-dependent modes on three scenes ----------------------------------------------------------------------
def profiled():
    """a profiler is preloaded (rocprofv3 sets ROCP* / ROCPROF* variables): every dispatch becomes a row of its output,
    so
    the legs keep their untimed loops short"""
    return any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ)


def warm_clocks(ctx, step, seconds=0.3):
    """A leg that prepares its inputs on the CPU for a second or two leaves the GPU idle, its clocks drop, and the first
    ~25 sweeps afterwards run 3 % longer while they ramp up again (profiles/variance_probe4.py: 1.092 instead of 1.061
    ms
    after 2 s of idle).  So such a leg runs its own step loop untimed for a moment before it times anything."""
    t_end = time.perf_counter() + (0.01 if profiled() else seconds)
    k = 0
    while time.perf_counter() < t_end:
        step(k)
        k += 1
        if k % 32 == 0:
            ctx.sync()
    ctx.sync()


def measure_modes(rig, ctx, frames, steps, warmup):
    """full sweep, pre_* chain, brick-skipping mode, background skip and store elision of one context fed `frames`
    (a list of (depth, colour) device tensors) in rotation"""
    k = [0]

    def step(bricks):
        d, c = frames[k[0] % len(frames)]
        k[0] += 1
        ctx.update_device(d.data_ptr(), c.data_ptr())
        ctx.clear_occupied_bricks()
        ctx.process_textures()
        ctx.update_occupied_bricks()
        ctx.integrate()

    res = {"frames_in_rotation": len(frames)}
    ctx.set_use_bricks(False)
    warm_clocks(ctx, step)
    dt, st = rig.timed(False, steps, warmup, step=step, ctx=ctx)
    int_ms = st["2integrate"][0] / max(st["2integrate"][1], 1) * 1e-6
    # the headline's roofline arithmetic on THIS input (bench.headline_line): algorithmic bytes of the launch over its
    # duration against the 8 TB/s peak, and against this context's own stream replay (rgbdr_settle: the kernel's memory
    # streams over its arena without arithmetic) -- equal across the scenes within the run's noise, or the kernel is
    # data dependent
    g = ctx.geo
    V = g.res_volume[0] * g.res_volume[1] * (g.slab_voxel_z1 - g.slab_voxel_z0)
    bytes_launch = V * (4 + 12 * rig.N) + rig.N * rig.W * rig.H * 8
    replay_ms = ctx.settle(0.0)
    res["full_sweep"] = {"ms_per_step": round(dt / steps * 1e3, 4), "integrate_ms": round(int_ms, 4),
                         "roofline_frac": round(bytes_launch / (int_ms * 1e-3) / 8e12, 4) if int_ms > 0 else None,
                         "box_stream_replay_ms": round(replay_ms, 4),
                         "frac_of_box_stream": round((bytes_launch / int_ms) / (V * (4 + 12 * rig.N) / replay_ms), 4)
                         if int_ms > 0 and replay_ms > 0 else None}
    _, st = rig.timed(False, max(4, len(frames)), 1, detail=1, step=step, ctx=ctx)
    res["pre_chain_ms"] = round(st["1preprocess"][0] / max(st["1preprocess"][1], 1) * 1e-6, 4)
    _, st = rig.timed(False, max(4, len(frames)), 1, detail=2, step=step, ctx=ctx)
    res["pre_passes_ms"] = {n: round(st[n][0] / max(st[n][1], 1) * 1e-6, 4) for n in ("morph", "bilateral", "boundary",
        "normal", "quality")}
    dt, st = rig.timed(True, steps, warmup, step=step, ctx=ctx)
    res["bricked"] = {"ms_per_step": round(dt / steps * 1e3, 4),
        "integrate_ms": round(st["2integrate"][0] / max(st["2integrate"][1], 1) * 1e-6, 4),
                      "occupied_ratio": round(ctx.occupied_ratio(), 4)}
    ctx.set_use_bricks(False)
    ctx.set_skip_background(True)
    try:
        dt, st = rig.timed(False, steps, warmup, step=step, ctx=ctx)
        sk = skip_summary(rig, ctx, dt, st, steps)
    finally:
        ctx.set_skip_background(False)
    res["background_skip"] = {key: sk[key] for key in ("ms_per_step", "integrate_ms", "frac_decided", "tiles_listed",
        "tiles")}
    ctx.set_elide_stores(True)
    try:
        dt, st = rig.timed(False, steps, warmup, step=step, ctx=ctx)
    finally:
        ctx.set_elide_stores(False)
    res["store_elision"] = {"ms_per_step": round(dt / steps * 1e3, 4),
        "integrate_ms": round(st["2integrate"][0] / max(st["2integrate"][1], 1) * 1e-6, 4)}
    res["valid_pixels"] = None
    return res


def leg_scenes(rig):
    """The data-dependent numbers on more than their best case: `static` is the headline's scene (two thirds
    of
    the pixels see nothing, the same frame every step); `moving` rotates four different frames of it (new noise and
    holes,
    the sphere displaced: occupied bricks, tile states, list sizes and elided stores change every step); `dense` is a
    scene whose every pixel is valid and inside the box (pre_* runs its 169 taps everywhere, nothing is background);
    `dense_moving` rotates four frames of that."""
    torch, capi, synth, ctx = rig.torch, rig.capi, rig.synth, rig.ctx
    steps, warmup = max(8, min(rig.args.steps, 24)), 4
    N, W, H = rig.N, rig.W, rig.H
    res = {}

    def resident(scene, count):
        frames = []
        for k in range(count):
            d, c = (scene.depth, scene.color) if k == 0 else scene.frame(k)
            frames.append((torch.from_numpy(d).to(rig.dev), torch.from_numpy(c).to(rig.dev)))
        return frames

    def valid(frames):
        return round(float(sum((f[0] > 0).float().mean().item() for f in frames) / len(frames)), 4)

    ring = resident(rig.scene, 4)
    res["static"] = measure_modes(rig, ctx, ring[:1], steps, warmup)
    res["static"]["valid_pixels"] = valid(ring[:1])
    res["moving"] = measure_modes(rig, ctx, ring, steps, warmup)
    res["moving"]["valid_pixels"] = valid(ring)
    # the dense scene has its own sensor poses, hence its own calibration: a second context, its LUT arena placed by the
    # same policy as the headline's (the trials bench.py asked for, or the library's default)
    dense_scene = synth.Scene(N, W, H, lut_res=(128, 106, 128), seed=1234, layout="dense")
    dctx = None
    try:
        cfg = capi.make_config(N, (W, H), voxel_size=2.0 / rig.G, brick_size=8 * 2.0 / rig.G, res_override=rig.grid)
        dctx = capi.Context(cfg, rig.local_rank)
        for i in range(N):
            dctx.set_calibration(i, dense_scene.xyz[i], dense_scene.lut_res, dense_scene.uv[i], dense_scene.lut_res,
                (0.5, 4.5))
            dctx.synth_inverse_calibration(i, dense_scene.pinhole(i))
        dense = resident(dense_scene, 4)
        res["dense"] = measure_modes(rig, dctx, dense[:1], steps, warmup)
        res["dense"]["valid_pixels"] = valid(dense[:1])
        res["dense_moving"] = measure_modes(rig, dctx, dense, steps, warmup)
        res["dense_moving"]["valid_pixels"] = valid(dense)
    finally:
        if dctx is not None:
            dctx.close()
    fr = [res[k]["full_sweep"]["frac_of_box_stream"] for k in ("static", "moving", "dense", "dense_moving") if k in res]
    res["full_sweep_frac_of_box_stream_spread"] = round(max(fr) - min(fr), 4) if fr and None not in fr else None
    res["note"] = ("ms per step of the same 4-sensor 512^3 job; `static` is the best case every other key of this "
        "line is "
                   "quoted on, `dense` / `dense_moving` bound the pre_* chain and the skipping modes from above; "
                   "full_sweep."
                   "roofline_frac is the headline's arithmetic on each input, frac_of_box_stream holds it against the "
                   "context's own stream replay (static / moving share the headline's arena, the dense pair has its "
                   "own, "
                   "placed by the same policy)")
    return res


# ---- consumers and producers either side of the path ---------------------------------------------------------------
def leg_post_pass(rig):
    """consumer of the volume (BASELINE configs[4] names the post-pass): ray-march at 720p with and without space
    skipping, hole filling"""
    ctx, capi, synth = rig.ctx, rig.capi, rig.synth
    ctx.set_use_bricks(False)
    ctx.integrate()
    ctx.set_timer_detail(2)
    ctx.enable_timers(True)
    try:
        view = capi.make_view((2.4, 1.8, 2.1), (0.0, 0.7, 0.0), (0.0, 1.0, 0.0), 45.0, 1280, 720, synth.BBOX_MIN,
            synth.BBOX_MAX)
        warm_clocks(ctx, lambda k: rig.step(False))
        ctx.raymarch(view)
        _, depth_img, _ = ctx.raymarch(view)
        ctx.fill_colors(1280, 720)
        ctx.fill_colors(1280, 720)
        full_ms = ctx.timer_ns("draw") * 1e-6
        view.skip_space = 1                 # brick depth peels -> start positions (reference default)
        ctx.raymarch(view)
        ctx.raymarch(view)
        skip_ms, peel_ms = ctx.timer_ns("draw") * 1e-6, ctx.timer_ns("brickdraw") * 1e-6
        return {"viewport": [1280, 720], "raymarch_ms": round(full_ms, 4),
                "raymarch_skip_space_ms": round(skip_ms, 4),        # the march alone, from the peeled start positions
                "brickdraw_ms": round(peel_ms, 4),                  # the depth peels before it
                "view_pass_skip_space_ms": round(skip_ms + peel_ms, 4),
                "holefill_ms": round(ctx.timer_ns("holefill") * 1e-6, 4),
                "surface_pixels": round(float((depth_img < 1).mean()), 4)}
    finally:
        ctx.enable_timers(False)


def leg_post_pass_slabs(rig):
    """post-pass across the slabs (BASELINE configs[4]): slab ray-march (find, all-reduce MIN, shade, composite) +
    tsdf_inpaint / tsdf_colorfill of the composited frame; --loopback / --slab run it too (one slab's share of the
    frame)"""
    ctx, capi, synth, rdist = rig.ctx, rig.capi, rig.synth, rig.rdist
    ctx.set_use_bricks(False)
    rig.step(False)
    rig.exchanger.wait()
    rig.barrier()
    view = capi.make_view((2.4, 1.8, 2.1), (0.0, 0.7, 0.0), (0.0, 1.0, 0.0), 45.0, 1280, 720, synth.BBOX_MIN,
        synth.BBOX_MAX)
    vh = rig.transport["kind"] != "rccl"
    group = rig.transport["group"]
    rdist.raymarch_slabs(ctx, view, rig.dev, group=group, via_host=vh)
    rig.barrier()
    t0 = time.perf_counter()
    for _ in range(3):
        col, dep, _ = rdist.raymarch_slabs(ctx, view, rig.dev, group=group, via_host=vh)
    rig.barrier()
    t_march = (time.perf_counter() - t0) / 3 * 1e3
    ctx.set_timer_detail(2)
    ctx.enable_timers(True)
    try:
        ctx.upload_view_frame(col.cpu().numpy(), dep.cpu().numpy())
        ctx.fill_colors(1280, 720)
        ctx.fill_colors(1280, 720)
        return {"viewport": [1280, 720], "slab_raymarch_composited_ms": round(t_march, 4),
                "holefill_ms": round(ctx.timer_ns("holefill") * 1e-6, 4),
                "surface_pixels": round(float((dep < 1).float().mean()), 4)}
    finally:
        ctx.enable_timers(False)


def leg_host_fed(rig):
    """the same step fed from HOST buffers (never part of `value`): pageable upload, the library's page-locked frame
    buffer with and without the producer's memcpy, and that under RGBDR_FLAG_PIPELINE"""
    ctx, scene, args = rig.ctx, rig.scene, rig.args

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
    warm_clocks(ctx, lambda k: rig.step(False))
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
    res = {"bytes_per_frame": int(depth_h.nbytes + color_h.nbytes),
           "ms_per_step_pageable_upload": round(pageable, 4),
           "ms_per_step_mapped_buffer_incl_fill": round(fed(mapped_fill), 4),
           "ms_per_step_mapped_buffer": round(fed(mapped_only), 4)}
    if not args.pipeline:       # RGBDR_FLAG_PIPELINE: upload + pre_* of frame k+1 overlap integrate of frame k
        ctx.set_pipelined(True)
        try:
            res["ms_per_step_mapped_buffer_pipelined"] = round(fed(mapped_only), 4)
        finally:
            ctx.set_pipelined(False)
    return res


def leg_reference_defaults(rig):
    """The reference's own default operating point: voxel 0.01 m over (-1,0,-1)-(1,2.2,1) -> 200 x 221 x 200, bricks of
    0.1 m (10 voxels), inverse LUTs at the calib_inverter default spacing 0.007 m (286 x 315 x 286, generated on the
    device, resampled to the grid at upload), DXT1 colour frames, 1280 x 1080 colour next to 512 x 424 depth,
    brick-skipping sweep.  Timed on the static ring scene and on four of its frames in rotation."""
    torch, capi, synth = rig.torch, rig.capi, rig.synth
    N, W, H = rig.N, rig.W, rig.H
    bmax = (1.0, 2.2, 1.0)
    sc = synth.Scene(N, W, H, lut_res=(128, 106, 128), seed=1234, color_wh=(1280, 1080))
    rc = capi.Context(capi.make_config(N, (W, H), color_wh=(1280, 1080), bbox_max=bmax, voxel_size=0.01, brick_size=0.1,
                                       compress_rgb=1), rig.local_rank)
    try:
        t0 = time.perf_counter()
        for i in range(N):
            rc.set_calibration(i, sc.xyz[i], sc.lut_res, sc.uv[i], sc.lut_res, (0.5, 4.5))
            rc.set_inverse_calibration(i, rc.generate_inverse_lut(i, (286, 315, 286)), (286, 315, 286))
        rc.sync()
        t_lut = time.perf_counter() - t0
        frames = []
        for k in range(4):
            d, c = (sc.depth, sc.color) if k == 0 else sc.frame(k)
            blocks = np.stack([synth.encode_dxt(c[i], 1) for i in range(N)])
            frames.append((torch.from_numpy(d).to(rig.dev), torch.from_numpy(np.ascontiguousarray(blocks)).to(rig.dev)))
        torch.cuda.synchronize()

        def run(count, n=100):
            def rstep(k):
                d, b = frames[k % count]
                rc.update_device(d.data_ptr(), b.data_ptr())
                rc.clear_occupied_bricks(); rc.process_textures(); rc.update_occupied_bricks(); rc.integrate()
            for k in range(8):
                rstep(k)
            rc.sync()
            t0 = time.perf_counter()
            for k in range(n):
                rstep(k)
            rc.sync()
            return (time.perf_counter() - t0) / n * 1e3

        def rs(k):
            d, b = frames[0]
            rc.update_device(d.data_ptr(), b.data_ptr())
            rc.clear_occupied_bricks(); rc.process_textures(); rc.update_occupied_bricks(); rc.integrate()
        warm_clocks(rc, rs)               # (the DXT encoding above kept the GPU idle for seconds)
        ms = run(1)
        occ = rc.occupied_ratio()
        ms_moving = run(4)
        return {"grid": list(rc.geo.res_volume), "brick_voxels": int(rc.geo.brick_voxels),
                "inverse_lut": [286, 315, 286], "colour": "DXT1 1280x1080",
                "ms_per_frame": round(ms, 4), "frames_per_s": round(1e3 / ms, 1),
                "ms_per_frame_moving": round(ms_moving, 4), "frames_per_s_moving": round(1e3 / ms_moving, 1),
                "occupied_ratio": round(occ, 4),
                "inverse_luts_generated_and_resampled_s": round(t_lut, 3)}
    finally:
        rc.close()


def leg_default_display_frame(rig):
    """ONE number for what the reference shows per frame in its default mode (source/kinect_client.cpp:572-617:
    NetKinectArray::update, clearOccupiedBricks, processTextures, updateOccupiedBricks, integrate, drawF =
    drawDepthLimits + ray-march + fillColors; bricks on, skip-space on, colorfill on, DXT1 1280 x 1080 colour, a
    1280 x 720 window), end to end on one stream with no host synchronisation inside the frame (rgbdr_draw): on the
    reference's own box (200 x 221 x 200 voxels, 10-voxel bricks, inverse LUT 286 x 315 x 286) and on the 512^3
    benchmark grid, static and with four frames in rotation, with the stages' event timers of one more frame."""
    torch, capi, synth = rig.torch, rig.capi, rig.synth
    N, W, H = rig.N, rig.W, rig.H
    sc = synth.Scene(N, W, H, lut_res=(128, 106, 128), seed=1234, color_wh=(1280, 1080))
    frames = []
    for k in range(4):
        d, c = (sc.depth, sc.color) if k == 0 else sc.frame(k)
        blocks = np.stack([synth.encode_dxt(c[i], 1) for i in range(N)])
        frames.append((torch.from_numpy(d).to(rig.dev), torch.from_numpy(np.ascontiguousarray(blocks)).to(rig.dev)))
    torch.cuda.synchronize()
    out = {"window": [1280, 720], "colour": "DXT1 1280x1080",
        "mode": "bricks on, skip-space on, colorfill on (the reference's defaults)"}

    def measure(rc, bbox_max):
        view = capi.make_view((2.4, 1.8, 2.1), (0.0, 0.7, 0.0), (0.0, 1.0, 0.0), 45.0, 1280, 720, synth.BBOX_MIN,
            bbox_max)
        view.skip_space = 1

        def frame(k, count):
            d, b = frames[k % count]
            rc.update_device(d.data_ptr(), b.data_ptr())
            rc.clear_occupied_bricks(); rc.process_textures(); rc.update_occupied_bricks(); rc.integrate()
            rc.draw(view, True)

        def run(count, n=100):
            for k in range(8):
                frame(k, count)
            rc.sync()
            t0 = time.perf_counter()
            for k in range(n):
                frame(k, count)
            rc.sync()
            return (time.perf_counter() - t0) / n * 1e3
        warm_clocks(rc, lambda k: frame(k, 1))
        res = {"grid": list(rc.geo.res_volume), "brick_voxels": int(rc.geo.brick_voxels)}
        res["ms_per_frame"] = round(run(1), 4)
        res["ms_per_frame_moving"] = round(run(4), 4)
        res["frames_per_s"] = round(1e3 / res["ms_per_frame"], 1)
        # the same frames with RGBDR_FLAG_PIPELINE: the pre_* chain of frame k + 1 on the second stream under the view
        # pass of frame k (throughput of a host that keeps enqueuing; a frame's latency is the sequential figure)
        rc.set_pipelined(True)
        try:
            res["ms_per_frame_pipelined"] = round(run(4), 4)
        finally:
            rc.set_pipelined(False)
        res["occupied_ratio"] = round(rc.occupied_ratio(), 4)
        rc.set_timer_detail(1)
        rc.enable_timers(True)
        try:
            frame(0, 1)
            frame(1, 1)
            rc.sync()
            res["stages_ms"] = {name: round(rc.timer_ns(t) * 1e-6, 4) for name, t in
                                (("pre_chain", "1preprocess"), ("integrate", "2integrate"), ("depth_peels",
                                "brickdraw"),
                                 ("raymarch", "draw"), ("holefill", "holefill"), ("drawF", "3recon"))}
        finally:
            rc.enable_timers(False)
            rc.set_timer_detail(0)
        _, dep = rc.readback_view_frame(True)
        res["surface_pixels"] = round(float((dep < 1).mean()), 4)
        return res

    bmax = (1.0, 2.2, 1.0)
    rc = capi.Context(capi.make_config(N, (W, H), color_wh=(1280, 1080), bbox_max=bmax, voxel_size=0.01, brick_size=0.1,
                                       compress_rgb=1), rig.local_rank)
    try:
        for i in range(N):
            rc.set_calibration(i, sc.xyz[i], sc.lut_res, sc.uv[i], sc.lut_res, (0.5, 4.5))
            rc.set_inverse_calibration(i, rc.generate_inverse_lut(i, (286, 315, 286)), (286, 315, 286))
        out["reference_box"] = measure(rc, bmax)
    finally:
        rc.close()
    keep = os.environ.get("RGBDR_ARENA_TRIALS")
    # (the brick-skipping sweep reads 3 % of the arena: placement does not matter)
    os.environ["RGBDR_ARENA_TRIALS"] = "1"
    rc = None
    try:
        rc = capi.Context(capi.make_config(N, (W, H), color_wh=(1280, 1080), voxel_size=2.0 / rig.G,
            brick_size=8 * 2.0 / rig.G,
                                           compress_rgb=1, res_override=rig.grid), rig.local_rank)
        for i in range(N):
            rc.set_calibration(i, sc.xyz[i], sc.lut_res, sc.uv[i], sc.lut_res, (0.5, 4.5))
            rc.synth_inverse_calibration(i, sc.pinhole(i))
        out["grid_512"] = measure(rc, synth.BBOX_MAX)
    finally:
        if keep is None:
            os.environ.pop("RGBDR_ARENA_TRIALS", None)
        else:
            os.environ["RGBDR_ARENA_TRIALS"] = keep
        if rc is not None:
            rc.close()
    return out


def leg_inverse_lut(rig):
    """f-3: the calib_inverter search (framework/calibration/calibration_inverter.cpp:99-155) on the device, at the
    benchmark grid: one sensor's 512^3 inverse LUT from its 128 x 106 x 128 forward LUT, straight into the resident
    layout.
    Its own context (it overwrites that sensor's LUT)."""
    capi = rig.capi
    G = rig.G
    keep = os.environ.get("RGBDR_ARENA_TRIALS")
    os.environ["RGBDR_ARENA_TRIALS"] = "1"
    c = None
    try:
        c = capi.Context(capi.make_config(1, (rig.W, rig.H), voxel_size=2.0 / G, brick_size=8 * 2.0 / G),
            rig.local_rank)
        c.set_calibration(0, rig.scene.xyz[0], rig.scene.lut_res, rig.scene.uv[0], rig.scene.lut_res, (0.5, 4.5))
        for _ in range(3):
            c.compute_inverse_calibration(0)     # (warm-up: the context above was built with the GPU idle)
        c.sync()
        times = []
        for _ in range(3):
            t0 = time.perf_counter()
            c.compute_inverse_calibration(0)
            c.sync()
            times.append(time.perf_counter() - t0)
        t = sorted(times)[1]
        return {"grid": [G, G, G], "forward_lut": list(rig.scene.lut_res), "inverse_lut_generate_ms": round(t * 1e3, 2),
                "Gvoxels_per_s": round(G ** 3 / t / 1e9, 2),
                "window": "library default (exact: widened until certified)",
                "what": "rgbdr_compute_inverse_calibration of one sensor, median of 3 after one warm-up"}
    finally:
        if keep is None:
            os.environ.pop("RGBDR_ARENA_TRIALS", None)
        else:
            os.environ["RGBDR_ARENA_TRIALS"] = keep
        if c is not None:
            c.close()


# ---- CPU baseline and the comparisons with the reference's own shaders ----------------------------------------------
def load_oracle():
    sys.path.insert(0, ROOT)
    from __graft_entry__ import load_oracle as lo
    return lo()


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


def median(v):
    v = sorted(v)
    return v[len(v) // 2]


def cpu_baseline(rig, reps=5):
    """Times the CPU oracle ("port") on this box's host cores over the benchmark workload itself: the whole
    pre_* chain for the frame set (median of `reps` runs after one warm-up) plus integrate() of the volume, taken in
    chunks
    of 64 z rows (the 1:1 LUT rows of a chunk are read back from the device first, untimed): every chunk is run once as
    warm-up -- that run is also compared with the HIP TSDF bit for bit, so a default bench run is a full-volume parity
    check at the benchmark size -- and then `reps` times; the chunk's time is the median of those.  --cpu-rows bounds
    the
    sample to that many rows mid-volume and extrapolates."""
    ctx, scene, capi, synth = rig.ctx, rig.scene, rig.capi, rig.synth
    N, rows, V_total = rig.N, rig.args.cpu_rows, rig.V_total
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
    for z0 in range(z_first, z_first + total_rows, chunk):
        n = min(chunk, z_first + total_rows - z0)
        inv = [ctx.readback_inverse_calibration(i, z0, z0 + n) for i in range(N)]
        res = (g.res_volume[0], g.res_volume[1], n)
        ref = orc.integrate(inv, sil, db, q, res, 0.01)               # warm-up of this chunk + the parity check
        got = hip[z0:z0 + n]
        parity = parity and bool(np.all((ref == got) | (np.isnan(ref) & np.isnan(got))))
        times = []
        for _ in range(reps):
            t0 = time.perf_counter()
            orc.integrate(inv, sil, db, q, res, 0.01)
            times.append(time.perf_counter() - t0)
        t_int += median(times)
        done += n

    def chain():
        orc.run_pipeline(scene, synth.BBOX_MIN, synth.BBOX_MAX, tuple(g.res_volume), None, brick_size=g.brick_size,
                         bv=tuple(g.brick_voxels_axis), res_bricks=tuple(g.res_bricks))
    chain()
    times = []
    for _ in range(reps):
        t0 = time.perf_counter()
        chain()
        times.append(time.perf_counter() - t0)
    t_pre = median(times)
    t_full = t_pre + t_int * (Z / done)
    res = {"value": round(V_total / t_full / 1e6, 2), "unit": "Mvoxels/s", "cores": threads,
           # the box's CPU and its logical CPU count
           "cpu_model": cpu_model(), "nproc": os.cpu_count(),
           "kind": "port", "repetitions": reps,
           "sample": "oracle (OpenMP, %d threads = CPUs granted by affinity and cgroup quota) on the benchmark "
           "workload: full "
                     "pre_* chain of the %d-sensor frame (median of %d after one warm-up: %.3f s) + integrate of %s "
                     "in chunks of 64 rows, "
                     "each chunk the median of %d runs after one warm-up (sum %.2f s)"
                     % (threads, N, reps, t_pre,
                     "all %d z rows" % Z if done == Z else "%d of %d z rows mid-volume, extrapolated to the "
                     "grid" % (done, Z),
                        reps, t_int),
           "integrate_mvoxels_per_s": round(g.res_volume[0] * g.res_volume[1] * done / t_int / 1e6, 2),
           "parity_rows_bit_exact": parity, "parity_rows": done}
    # comparisons with the reference's own shaders: each its own guarded sub-leg (they are not the baseline)
    for key, fn in (("reference_cpu_work", lambda: reference_cpu_work(rig, orc)),
                    ("reference_shader_text", lambda: reference_text_baseline(rig, hip, sil, db, q)),
                    ("reference_glsl_on_mesa", lambda: reference_glsl_checks(rig)),
                    ("driver_weight_bound", lambda: driver_weight_bound())):
        try:
            res[key] = fn()
        except Exception as e:  # noqa: BLE001
            res[key] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
    return res


def driver_weight_bound():
    """What moves when every LINEAR weight is held with 8 fractional bits, as the authors' NVIDIA driver does
    (INTEGRATION.md
    section 6): the oracle with exact against the oracle with 8-bit weights on four 512 x 424 sensors into 128^3 (the
    Mesa
    sample's scene), ~6 s of CPU.  The bound a maintainer comparing against a real NVIDIA run should expect -- derived,
    not
    a parity claim of the HIP path."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import driver_weight_bound as dwb
    c = dwb.case_sample()
    t = c["tsdf"]
    return {"what": "oracle, exact vs 8-bit LINEAR weights (GL 4.4 8.14; CalibVolumes.cpp:76,135,140), " + c["what"],
            "tsdf_voxels_in_band": t["voxels_in_band"], "tsdf_median_abs_diff_in_band": t["median_abs_diff_in_band"],
            "tsdf_p99_abs_diff_in_band": t["p99_abs_diff_in_band"], "tsdf_max_abs_diff": t["max_abs_diff"],
            "tsdf_voxels_beyond_5e-7": t["voxels_beyond_5e-7"], "tsdf_voxels_beyond_1e-4": t["voxels_beyond_1e-4"],
            "tsdf_voxels_changing_class": t["voxels_changing_class"],
            "depth_texels_flipping_validity": c["depth_rg"]["values_differing"],
            "brick_increments_moving": c["brick_counters"]["sum_abs_diff"],
            "brick_increments": c["brick_counters"]["sum"],
            "occupied_bricks_on_one_side_only": c["occupied_list"]["only_exact"] + c["occupied_list"]["only_8bit"],
            "all_cases": "profiles/r06_driver_weight_bound.json"}


def reference_cpu_work(rig, orc):
    """the reference's own per-resize / per-frame CPU work on this path, single-threaded as in the reference"""
    g = rig.ctx.geo
    try:
        return orc.reference_cpu_work(rig.synth.BBOX_MIN, rig.synth.BBOX_MAX, tuple(g.res_volume), g.brick_size,
                                      rig.ctx.readback_brick_counters(), 10)
    except MemoryError:
        return None


def reference_glsl_checks(rig):
    capi, synth = rig.capi, rig.synth
    ref_glsl = reference_glsl_fixture_check(capi, synth)
    if isinstance(ref_glsl, dict) and "error" not in ref_glsl:
        ref_glsl["baseline_sensor_size"] = reference_glsl_sample_check(capi, synth)
        ref_glsl["default_mode_bricks_on"] = reference_glsl_mode_check(capi, synth)
        ref_glsl["headline_grid_z_bands"] = reference_glsl_big_check(capi, synth, "four_sensors_512x424_into_512_bands")
        ref_glsl["default_mode_dxt1_bricks_at_sensor_size"] = reference_glsl_big_check(capi, synth,
            "default_mode_dxt1_bricks_512x424_into_128")
    return ref_glsl


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
        return {"what": "HIP path vs the reference's GLSL run on Mesa llvmpipe (committed fixture "
            "gl_passes_%s.npz)" % name,
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
    flips = (cls(t) != cls(r)) & ok
    # a voxel whose two values lie within 1e-6 of the SAME boundary: a last-bit difference of sdist against +-limit
    # (tsdf_integration.vs:41-46) turns exactly -limit into a weighted mean a hair above it -- tests/test_gl_ref.py
    # class_flips
    tie = (np.abs(np.abs(t) - lim) <= 1e-6) & (np.abs(np.abs(r) - lim) <= 1e-6) & (np.sign(t) == np.sign(r))
    return {"tsdf_max_abs_diff": float("%.3g" % (d.max() if d.size else 0.0)),
        "tsdf_voxels_beyond_5e-7": int((d > 5e-7).sum()),
            "tsdf_voxels_compared": int(ok.sum()), "tsdf_voxels_in_band": int((np.abs(r[ok]) < lim).sum()),
            "tsdf_voxels_changing_class": int(flips.sum()),
            "tsdf_voxels_changing_class_at_a_boundary_tie": int((flips & tie).sum())}


def reference_glsl_mode_check(capi, synth, name="bricks_reference_box_5_voxel_bricks"):
    """The library in the reference's DEFAULT mode (bricks on) against the Mesa run of the same mode: the reference's
    own
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
        out = {"what": "HIP path with RGBDR_FLAG_USE_BRICKS vs the reference's GLSL drawn through the occupied "
            "bricks' index lists on "
                       "Mesa (gl_passes_%s.npz: grid %s, %d of %d bricks occupied)" % (name,
                       "x".join(str(v) for v in geo.res_volume),
                                                                                       fx["occupied"].size,
                                                                                       fx["counters"].size),
               "brick_counters_equal": bool(np.array_equal(ctx.readback_brick_counters(), fx["counters"])),
               "occupied_bricks_equal": bool(np.array_equal(ctx.get_occupied()[0], fx["occupied"]))}
        out.update(_tsdf_summary(ctx.readback_tsdf(), fx["tsdf"], cfg.tsdf_limit))
        ctx.close()
        return out
    except Exception as e:  # noqa: BLE001 -- an extra key must never cost the line
        return {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}


def reference_glsl_big_check(capi, synth, name):
    """The larger Mesa samples of tests/golden/make_gl_golden.py BIG_SAMPLES: z bands of the 512^3 HEADLINE grid from
    four
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
            got = np.stack([ctx.readback_image(which, i) for i in range(n)]).reshape(n * H * W,
                -1)[tex].astype(np.float64)
            want = fx[k].astype(np.float64)
            fin = np.isfinite(got) & np.isfinite(want)
            imgs[k] = float("%.3g" % np.abs(got - want)[fin].max())
        out = {"what": "%s: %d sampled texels per image (every edge-class texel of pre_boundary among them), %d "
            "sampled voxels" % (
                   name, tex.size, fx["voxels"].size),
               "max_abs_diff": imgs,
               "brick_counts_differing": int(np.abs(ctx.readback_brick_counters().astype(np.int64)
                                                    - fx["counters"].astype(np.int64)).sum()),
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
    """the same at BASELINE's sensor size: four 512 x 424 sensors into 128^3, against the committed SAMPLE of the Mesa
    run
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
        for k, which in {"morph": 1, "depth_rg": 2, "lab": 3, "depth_b": 4, "sil": 5, "normal": 6,
            "quality": 7}.items():
            got = np.stack([c.readback_image(which, i) for i in range(n)]).reshape(n * H * W,
                -1)[tex].astype(np.float64)
            want = fx[k].astype(np.float64)
            fin = np.isfinite(got) & np.isfinite(want)
            out[k] = float("%.3g" % np.abs(got - want)[fin].max())
        cnt = c.readback_brick_counters().astype(np.int64)
        t = c.readback_tsdf().reshape(-1)[fx["voxels"].astype(np.int64)]
        c.close()
        r = fx["tsdf"]
        ok = ~(np.isnan(t) | np.isnan(r))
        lim = np.float32(cfg.tsdf_limit)
        return {"what": "4 sensors 512 x 424 into 128^3, %d sampled texels per image, %d sampled voxels (%d in the "
            "band)"
                        % (tex.size, t.size, int((np.abs(r[ok]) < lim).sum())),
                "max_abs_diff": out, "brick_counts_differing": int(np.abs(cnt - fx["counters"].astype(np.int64)).sum()),
                "brick_counts": int(fx["counters"].sum()),
                "tsdf_max_abs_diff": float("%.3g" % np.abs(t.astype(np.float64) - r)[ok].max()),
                "tsdf_voxels_beyond_1e-6": int((np.abs(t.astype(np.float64) - r)[ok] > 1e-6).sum())}
    except Exception as e:  # noqa: BLE001 -- an extra key must never cost the line
        return {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}


def reference_text_baseline(rig, hip, sil, db, q, rows=32):
    """The TEXT of the reference's own shaders compiled as C++ (oracle/_ref/libref_shaders.so, built in the build
    container by oracle/build_shader_ref.py; samplers are stand-ins, see oracle/glsl_runtime.hpp), one thread, on a
    bounded sample of the benchmark workload: the pre_* chain of sensor 0 and tsdf_integration.vs on `rows` z rows in
    the middle of the volume -- timed, and compared bit for bit with the HIP images / volume rows.  None where the
    library did not travel (it exists only where /root/reference was present at build time)."""
    ctx, scene, capi, synth, N = rig.ctx, rig.scene, rig.capi, rig.synth, rig.N
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
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
    same_img = all(bool(np.all((frame[k][0] == ctx.readback_image(w,
        0)) | (np.isnan(frame[k][0]) & np.isnan(ctx.readback_image(w, 0)))))
                   for k, w in (("depth_b", capi.IMG_DEPTH_B_RG), ("sil", capi.IMG_SILHOUETTE), ("quality",
                   capi.IMG_QUALITY),
                                ("normal", capi.IMG_NORMAL), ("lab", capi.IMG_LAB)))
    return {"what": "the reference's shader text compiled as C++ (stand-in samplers), 1 thread: tsdf_integration.vs "
        "on %d of %d "
                    "z rows, pre_* chain of 1 of %d sensors" % (rows, Z, N),
            "integrate_mvoxels_per_s": round(X * Y * rows / t_int / 1e6, 2), "integrate_s": round(t_int, 2),
            "pre_chain_one_sensor_s": round(t_pre, 2),
            "hip_rows_bit_identical": same_vol, "hip_images_bit_identical": same_img, "rows": rows}
