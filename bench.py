#!/usr/bin/env python3
"""bench.py -- TSDF fusion + depth preprocessing on MI355X: the headline measurement and the one JSON line.

A step is one pass of the hot path over one synthetic frame set already resident
in HBM: NetKinectArray::update (device->device copy of the resident frames),
clearOccupiedBricks, processTextures (morph/bilateral/boundary/normal/quality),
updateOccupiedBricks and a FULL-SWEEP integrate() of every voxel
(source/kinect_client.cpp:572-602).  Everything else the line carries (brick-skipping mode, other schedules,
dense / moving scenes, post-pass, host-fed frames, CPU baseline ...) is an extra key measured by bench_legs.py
AFTER the headline, each leg isolated: a leg that throws or hangs costs its own key, never the line.

N = 1: BASELINE.json configs[2] "4 sensors, 512^3 TSDF, full pre_* chain".
N > 1: one process per GPU, the volume is split into Z slabs of storage-tile
layers (no data-path collective for integration; the one exchange per step is the
halo tile layers to the Z neighbours over RCCL).  Headline: the N = 1 workload at fixed work per GPU
(4 sensors; 512^3 / 512x512x1024 / 512x1024x1024 / 1024^3 for 1 / 2 / 4 / 8 GPUs, "scaling": "weak"), so that
value(N) compares with N x value(1).  BASELINE.json's own multi-GPU configs (8 sensors: configs[3] at N = 2 / 4,
configs[4] at N = 8) are timed in the same run under baseline_configs_run; --baseline-configs swaps the roles.

An N > 1 run cannot end without a JSON line: every rank runs under a supervisor process that never touches the
GPU and walks a ladder of fresh child processes (RUNGS) inside a fixed time budget -- bench_launch.py.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from bench_chain import (all_ranks_ok, choose_chain, leave_lagged_chain, recheck_lagged_headline,  # noqa: E402,F401
                         retime_after_a_host_stall)
# noqa: E402,F401
from bench_launch import (EXIT_WATCHDOG, RUNG_BUDGETS, RUNGS, FileStore, Watchdog, budget_scale, free_port,
                          launch_ranks, supervise_rank)

# weak scaling: ~134 M voxels (one 512^3 worth) per GPU over the same 2 m box; every
# axis stays a power of two so the 1:1 inverse LUT needs no interpolation
GRID_FOR_GPUS = {1: (512, 512, 512), 2: (512, 512, 1024), 4: (512, 1024, 1024), 8: (1024, 1024, 1024)}
HBM_PEAK = 8.0e12  # B/s, MI355X_MICROARCH.md "HBM3E peak BW"

def choose_workload(world, loopback=False, weak=False, sensors=0, cubic_grid=0):
    """Which BASELINE.json config a run with `world` GPUs is: (sensors, grid, config string, scaling)."""
    if world == 1 or loopback or weak:
        n = sensors or 4
        grid = GRID_FOR_GPUS.get(world, (512, 512, 512))
        single = "configs[2]: 4 sensors, 512^3 TSDF, full pre_* depth-filter chain on 1 MI355X"
        cfg = single if world == 1 and not loopback \
            else ("configs[2] at fixed work per GPU (weak scaling): %d sensors, 512^3 voxels per MI355X as Z slabs of "
                "a %dx%dx%d TSDF, "
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


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--grid", type=int, default=0, help="override with a cubic grid of this size")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (gloo: debugging several "
                                                      "ranks on one GPU)")
    ap.add_argument("--sensors", type=int, default=0,
        help="0 = what the BASELINE config of --gpus names (4 at 1 GPU, 8 above)")
    ap.add_argument("--weak", action="store_true",
        help="N > 1: only the weak-scaling grids (4 sensors, 512^3 voxels per GPU), without the "
                                                        "BASELINE configs[3]/[4] run that the default adds under "
                                                        "baseline_configs_run")
    ap.add_argument("--baseline-configs", action="store_true",
                    help="N > 1: make BASELINE configs[3] (N = 2, 4) / configs[4] (N = 8) -- 8 sensors -- the "
                    "headline and report the "
                         "weak-scaling twin under weak_scaling_4_sensors (the default is the other way round: the "
                         "headline of an N > 1 "
                         "run is the N = 1 workload at fixed work per GPU, so that value(N) is comparable with N x "
                         "value(1))")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="headline only: none of the extra keys of bench_legs.py")
    ap.add_argument("--pipeline", action="store_true",
                    help="RGBDR_FLAG_PIPELINE for the headline: the pre_* chain of step k+1 overlaps integrate of "
                    "step k on a "
                         "second stream (2-3 %% more frames/s, but the integrate launches it is measured on run 4 %% "
                         "longer "
                         "under the overlap; the other schedule is always reported under 'other_schedule')")
    ap.add_argument("--loopback", action="store_true",
                    help="one GPU, real RCCL: run an inner Z slab (rank 1 of 4) whose two neighbours are this process "
                         "itself -- exercises the whole N > 1 code path (probe, staging, side stream); value is per "
                         "slab")
    ap.add_argument("--arena-trials", type=int, default=0,
                    help="RGBDR_ARENA_TRIALS for the headline context; 0 (default) = leave the library's own default "
                    "in "
                         "force (up to 16 candidate placements of the LUT arena for arenas of 1 GiB and more; an "
                         "RGBDR_ARENA_TRIALS "
                         "already in the environment is honoured).  Within one box the sweep time differs by up to 12 "
                         "%% "
                         "with where hipMalloc placed the arena; the candidates' times are reported in "
                         "roofline.arena_placement_probe_ms, next to what the first placement alone "
                         "(frac_first_placement) and "
                         "round 3's library default of three (frac_first_3) would have given")
    ap.add_argument("--cpu-rows", type=int, default=0,
                    help="bound the CPU baseline to this many z rows of the volume (0 = the whole volume, about 10 s)")
    ap.add_argument("--slab", default="",
                    help="r/k: one GPU, real RCCL: run Z slab r of k of BASELINE configs[3] (k = 2, 4: 8 sensors, "
                    "512^3) or "
                         "configs[4] (k = 8: 8 sensors, 1024^3) exactly as rank r of a --gpus k run would -- the "
                         "sweep that "
                         "stages its boundary layers, the exchange on the side stream with this process as its own "
                         "neighbour(s); value is this slab's share")
    ap.add_argument("--slab-sweep", type=int, default=0,
                    help="k: --slab r/k for every r in one process, one JSON line with a row per rank (projection of "
                    "the "
                         "multi-GPU balance from single-GPU runs; DESIGN.md section 6)")
    ap.add_argument("--no-shard", dest="shard", action="store_false",
                    help="N > 1: every rank runs the pre_* chain for ALL sensors (rounds 1-3) instead of its n / N "
                    "share followed by "
                         "the all-gather of the packed frames and the all-reduce of the brick counters "
                         "(rgbdr_set_sensor_shard; "
                         "measured per rank of configs[3]: 0.68 against 0.71-0.73 ms per frame, profiles/r04_notes)")
    ap.add_argument("--no-lagged", dest="lagged", action="store_false",
                    help="N > 1: do not try the lagged chain schedule (the fallback rungs of the launch ladder pass "
                    "this: "
                         "after a failed rung only the plain schedules are candidates)")
    ap.add_argument("--torch-collectives", dest="managed", action="store_false",
                    help="N > 1 over RCCL: halo exchange and frame gather through torch.distributed's process group "
                    "(rounds 1-3) "
                         "instead of the C ABI's managed forms, where the LIBRARY enqueues them on its own streams "
                         "with a raw RCCL "
                         "communicator (what a C++ host does, host/slab_loop.cpp); gloo runs always go through torch")
    ap.set_defaults(shard=True, managed=True)
    ap.add_argument("--launch-timeout", type=float, default=1500.0,
                    help="--gpus N > 1: seconds for the whole ladder (every rung together); the driver waits 1800")
    ap.add_argument("--rung-budgets", default="",
                    help="--gpus N > 1: seconds per rung of the ladder, comma separated (default "
                    "%s)" % ",".join("%d" % b for b in RUNG_BUDGETS))
    ap.add_argument("--halo-transport", choices=("rccl", "peer"), default="rccl",
                    help="N > 1 / --slab: what moves the halo faces -- RCCL send / recv (default, the transport "
                    "north_star names) or "
                         "'peer': the copy engine, every rank pulling its neighbours' staged faces from their "
                         "IPC-mapped staging sets "
                         "(rgbdr_halo_export / _set_peer / _pull_async; the gather of a sharded chain stays on RCCL)")
    ap.add_argument("--first-rung", type=int, default=0, help="--gpus N > 1: start the ladder at this rung (0-3)")
    return ap.parse_args(argv)


def main():
    args = parse_args()
    role = os.environ.get("RGBDR_BENCH_ROLE", "")
    # An N > 1 run: whoever started this process (the driver directly, torch.distributed.run, a shell), it does not
    # touch the GPU.  Without WORLD_SIZE it starts one supervisor per rank; a supervisor (WORLD_SIZE set by us or by
    # torch.distributed.run) walks the ladder with fresh children of role "rank", which do the GPU work.
    if args.gpus > 1 and role != "rank" and not args.slab_sweep:
        if "WORLD_SIZE" not in os.environ:
            sys.exit(launch_ranks(args.gpus, sys.argv[1:], timeout=args.launch_timeout))
        sys.exit(supervise_rank(args, sys.argv[1:]))
    # stdout carries the one JSON line and nothing else: RCCL's version banner, gloo's connection notes and any
    # other chatter of the libraries below go to stderr
    global EMIT
    sys.stdout.flush()
    EMIT = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    if args.slab_sweep:
        sys.exit(slab_sweep(args))
    # What an N > 1 run times.  Default: the N = 1 workload at fixed work per GPU -- 4 sensors, 512^3
    # voxels per
    # rank as Z slabs of one larger volume, halo exchange and all ("scaling": "weak") -- as the headline, so that
    # value(N)
    # compares with N x value(1); BASELINE.json's own multi-GPU configs (8 sensors: configs[3] at N = 2 / 4, configs[4]
    # at
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

# What stays on the ONE stdout line: the driver's fields, `roofline` and `cpu_baseline` (flat: the driver's record
# keeps the
# scalars of those two), and at most ten keys more.  Everything else a run measures -- the data-dependent modes per
# scene,
# host-fed frames, the reference's default operating point, the comparisons with the Mesa run in full, the per-rank
# tables of
# an N > 1 run -- goes to bench_extra.json beside this file (RGBDR_BENCH_EXTRA overrides the path; N > 1:
# bench_extra_nN.json).
LINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
    "vs_baseline",
             "dtype", "data", "config", "roofline", "cpu_baseline",
             "frames_per_s", "passes_ms", "default_display_frame", "post_pass", "halo", "launch", "legs_incomplete",
             "provisional",
             "value_first_measurement", "extra")


def flat_cpu_baseline(c):
    """cpu_baseline for the line: its scalars, plus the parity facts of the nested comparisons as scalars"""
    if not isinstance(c, dict):
        return c, None
    flat = {k: v for k, v in c.items() if not isinstance(v, (dict, list))}
    nested = {k: v for k, v in c.items() if isinstance(v, (dict, list))}

    def take(prefix, d, keys):
        if isinstance(d, dict):
            for k in keys:
                if k in d and not isinstance(d[k], (dict, list)):
                    flat[prefix + k] = d[k]
            if "error" in d:
                flat[prefix + "error"] = d["error"]
    rg = nested.get("reference_glsl_on_mesa")
    take("glsl_on_mesa_", rg, ("tsdf_max_abs_diff", "tsdf_voxels_changing_class", "brick_counters_equal",
        "voxels_masked_nan_on_llvmpipe_only"))
    if isinstance(rg, dict):
        take("glsl_on_mesa_sensor_size_", rg.get("baseline_sensor_size"), ("tsdf_max_abs_diff",
            "brick_counts_differing", "brick_counts"))
        take("glsl_on_mesa_512_bands_", rg.get("headline_grid_z_bands"),
             ("tsdf_max_abs_diff", "tsdf_voxels_beyond_5e-7", "tsdf_voxels_compared", "tsdf_voxels_changing_class",
             "brick_counts_differing"))
        take("glsl_on_mesa_default_mode_", rg.get("default_mode_dxt1_bricks_at_sensor_size"),
             ("tsdf_max_abs_diff", "tsdf_voxels_changing_class", "tsdf_voxels_changing_class_at_a_boundary_tie",
             "occupied_bricks_equal"))
    take("reference_shader_text_", nested.get("reference_shader_text"), ("hip_rows_bit_identical",
        "hip_images_bit_identical"))
    take("driver_weight_bound_", nested.get("driver_weight_bound"),
         ("tsdf_p99_abs_diff_in_band", "tsdf_median_abs_diff_in_band", "tsdf_voxels_changing_class",
         "depth_texels_flipping_validity",
          "brick_increments_moving", "occupied_bricks_on_one_side_only"))
    return flat, nested


def split_line(obj):
    """(the line, the extra object or None)"""
    if "roofline" not in obj or "value" not in obj:
        return obj, None                                   # an error line: as it is
    line = {k: obj[k] for k in LINE_KEYS if k in obj}
    extra = {k: v for k, v in obj.items() if k not in LINE_KEYS}
    flat, nested = flat_cpu_baseline(obj.get("cpu_baseline"))
    if flat is not None and "cpu_baseline" in obj:
        line["cpu_baseline"] = flat
        extra["cpu_baseline"] = nested
    roof = dict(obj["roofline"])
    box = roof.pop("box", None)                            # clocks / power under load: a nested table
    if box is not None:
        extra["roofline_box"] = box
        for k in ("sclk_MHz", "power_W"):
            if isinstance(box, dict) and isinstance(box.get(k), dict):
                roof["box_" + k + "_median"] = box[k].get("median")
    sc = obj.get("scenes")
    # is the headline's fraction a property of the kernel or of its best-case input?
    if isinstance(sc, dict):
        for name in ("static", "moving", "dense", "dense_moving"):
            fs = sc.get(name, {}).get("full_sweep") if isinstance(sc.get(name), dict) else None
            if isinstance(fs, dict):
                roof["frac_scene_" + name] = fs.get("roofline_frac")
                roof["frac_of_box_stream_scene_" + name] = fs.get("frac_of_box_stream")
    pr = obj.get("per_rank")
    # N > 1: the per-rank tables go to the file, their range stays
    if isinstance(pr, dict) and pr.get("roofline_frac"):
        fr = [f for f in pr["roofline_frac"] if f is not None]
        roof.update(ranks=len(pr["roofline_frac"]), frac_slowest_rank=min(fr) if fr else None,
            frac_fastest_rank=max(fr) if fr else None,
                    integrate_ms_slowest_rank=max(pr["integrate_ms"]),
                    integrate_ms_fastest_rank=min(pr["integrate_ms"]))
    # e.g. the replay time of every arena candidate
    tables = {k: v for k, v in roof.items() if isinstance(v, (dict, list))}
    if tables:
        extra["roofline_tables"] = tables
        roof = {k: v for k, v in roof.items() if k not in tables}
    line["roofline"] = roof
    # the two consumer keys: compact on the line, whole in the file
    for k in ("default_display_frame", "post_pass"):
        if isinstance(line.get(k), dict) and "error" not in line[k]:
            extra[k] = line[k]
    d = line.get("default_display_frame")
    if isinstance(d, dict) and "error" not in d:
        line["default_display_frame"] = {g: {"ms_per_frame": d[g].get("ms_per_frame"),
            "ms_per_frame_moving": d[g].get("ms_per_frame_moving"),
                                             "stages_ms": d[g].get("stages_ms")} for g in ("reference_box",
                                             "grid_512") if isinstance(d.get(g), dict)}
    return line, extra


def extra_path(obj):
    n = obj.get("n_gpus", 1)
    return os.environ.get("RGBDR_BENCH_EXTRA") or os.path.join(ROOT,
        "bench_extra.json" if n == 1 else "bench_extra_n%d.json" % n)


def emit(obj):
    line, extra = split_line(obj)
    if extra is not None:
        path = extra_path(obj)
        try:
            tmp = path + ".tmp%d" % os.getpid()
            with open(tmp, "w") as f:
                json.dump(extra, f, indent=1, sort_keys=True)
            os.replace(tmp, path)
            line["extra"] = {"file": path, "keys": sorted(extra)}
        except OSError as e:                               # a read-only checkout: the line alone
            line["extra"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:120]), "keys": sorted(extra)}
    EMIT.write(json.dumps(line) + "\n")
    EMIT.flush()


# ---------------------------------------------------------------------------------------------------------------------
# one rank
# ---------------------------------------------------------------------------------------------------------------------
def parse_slab(text):
    """'r/k' -> (r, k)"""
    try:
        r, k = (int(t) for t in text.split("/"))
    except ValueError:
        raise SystemExit("--slab takes r/k, e.g. 1/4")
    if k not in (2, 4, 8) or not 0 <= r < k:
        raise SystemExit("--slab r/k: k is 2 or 4 (configs[3]) or 8 (configs[4]), 0 <= r < k")
    return r, k


class Rig:
    """What the headline and every leg of bench_legs.py share: the context, the resident frames, the step and the
    timer."""
    HBM_PEAK = HBM_PEAK

    def __init__(self, args, slab, quiet, shared):
        self.args, self.quiet, self.shared = args, quiet, shared
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.supervised = os.environ.get("RGBDR_BENCH_ROLE") == "rank"
        if slab is None and args.slab:
            slab = parse_slab(args.slab)
        if slab is None and args.loopback:
            slab = (1, 4)                    # --loopback: an inner slab of configs[3]
        self.slab = slab
        self.loop = slab is not None         # one GPU stands in for rank r of k: it is its own neighbour(s)
        self.multi = self.world > 1 or self.loop
        self.out = None                      # the JSON object once the headline exists ...
        self.printed = False                 # ... and whether the line has left the process
        self.exchanger = self.gather = self.halo = self.transport = self.rccl_info = self.lag = None
        self.managed = False
        self.watchdog = shared.get("watchdog") or Watchdog(self.expired, tag=" rank %d" % self.rank)
        shared["watchdog"] = self.watchdog
        # (the twin run shares the thread; the line at stake is the newest rig's)
        self.watchdog.on_expire = self.expired

    def expired(self, name, budget):
        """the watchdog's last word: with a headline in hand rank 0 prints the line as far as it got, status 0"""
        top = self.shared.get("top_rig") or self
        if top.out is None or top.quiet:
            os._exit(EXIT_WATCHDOG)
        if self.rank == 0 and not top.printed:
            key = name[4:] if name.startswith("leg ") else name
            top.out.setdefault(key, {"error": "watchdog: exceeded %.0f s" % budget})
            top.out["legs_incomplete"] = "stopped by the watchdog in '%s'" % name
            top.out.pop("provisional", None)
            emit(top.out)
        os._exit(0)

    def step(self, bricks=False):
        ctx = self.ctx
        if self.lag is not None:             # dist.LaggedChain: chain + gather of this frame, sweep of the one before
            self.lag.push(self.d_depth.data_ptr(), self.d_color.data_ptr())
            return
        ctx.update_device(self.d_depth.data_ptr(), self.d_color.data_ptr())
        ctx.clear_occupied_bricks()
        ctx.process_textures()
        if self.gather is not None:
            # the other ranks' sensors: all-gather of the packed frames, all-reduce of the brick counts
            self.gather()
        ctx.update_occupied_bricks()
        if self.halo is not None:
            self.exchanger.begin_step()      # the sweep stores its boundary layers into a staging set
        ctx.integrate()
        if self.halo is not None:
            self.exchanger.exchange_async()

    def barrier(self):
        self.ctx.sync()
        self.torch.cuda.synchronize()
        if self.multi:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def timed(self, bricks, steps, warmup, detail=0, step=None, ctx=None):
        """`steps` steps between two barriers; returns (seconds -- the MAX over ranks, {timer: (ns, launches)})"""
        torch, dist = self.torch, self.dist
        ctx = ctx or self.ctx
        step = step or self.step
        ctx.set_use_bricks(bricks)
        for _ in range(warmup):
            step(bricks)
        self.barrier()
        # timed region: only the integrate launches carry HIP events (their duration is needed
        # for the roofline; each event record costs ~4 us of stream time); the per-pass breakdown
        # comes from a separate short run
        ctx.set_timer_detail(detail)
        ctx.enable_timer_accumulation(True)
        t0 = time.perf_counter()
        t_prev, longest = t0, 0.0
        for _ in range(steps):
            step(bricks)
            t_now = time.perf_counter()
            longest = max(longest, t_now - t_prev)
            t_prev = t_now
        self.host_enqueue_ms = (t_prev - t0) / steps * 1e3   # the host's share: enqueue time per step
        self.longest_host_step_ms = longest * 1e3            # (a host stall inside the collective library shows here)
        if ctx is not self.ctx:
            ctx.sync()
        self.barrier()
        dt = time.perf_counter() - t0
        names = ("2integrate",) + (("1preprocess", "bricks") if detail > 0 else ()) + \
                (("morph", "bilateral", "boundary", "normal", "quality") if detail > 1 else ())
        stats = {n: ctx.timer_stats(n) for n in names}
        ctx.enable_timer_accumulation(False)
        ctx.enable_timers(False)
        ctx.set_timer_detail(2)          # the library's default again (detail 0 mutes every timer but "2integrate")
        self.local_dt = dt
        if self.multi:
            t = torch.tensor([dt], dtype=torch.float64, device=self.dev if self.args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, stats

    def restore_defaults(self):
        """after a leg failed half-way: the switches a leg may have left on"""
        ctx = self.ctx
        for fn in (lambda: ctx.set_use_bricks(False), lambda: ctx.set_pipelined(bool(self.args.pipeline)),
                   lambda: ctx.set_elide_stores(False), lambda: ctx.set_skip_background(False),
                   lambda: ctx.enable_timer_accumulation(False), lambda: ctx.enable_timers(False),
                   lambda: ctx.set_timer_detail(2)):
            try:
                fn()
            except Exception:  # noqa: BLE001
                pass


def open_context(rig):
    """torch, the process group, the scene resident in HBM, the library context with its calibration"""
    from __graft_entry__ import load_package
    import torch
    import torch.distributed as dist
    args, shared, world, rank = rig.args, rig.shared, rig.world, rig.rank
    rig.torch, rig.dist = torch, dist
    if world != args.gpus and not (world == 1 and args.gpus <= 1):
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    if rig.loop and world != 1:
        raise SystemExit("--slab / --loopback run on one GPU (they stand in for a --gpus k run)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the hot path has no CPU fallback")
    if args.backend == "gloo":
        rig.local_rank = rig.local_rank % torch.cuda.device_count()
    torch.cuda.set_device(rig.local_rank)
    dev = rig.dev = torch.device("cuda", rig.local_rank)
    if rig.loop:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
    if rig.multi and not shared.get("pg"):
        import datetime
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # a collective that waits longer is a hang
        limit = datetime.timedelta(seconds=max(60.0, 240.0 * rig.watchdog.scale))
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=limit)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world, timeout=limit)
        shared["pg"] = True
    # several ranks on one GPU (--backend gloo, debugging): no placement shopping, it would hold world x 10 arenas
    if args.backend == "gloo" and world > 1:
        os.environ["RGBDR_ARENA_TRIALS"] = "1"
    elif args.arena_trials > 0:
        # read by the library when the LUT arena is created
        os.environ["RGBDR_ARENA_TRIALS"] = str(args.arena_trials)
    rig.trials = os.environ.get("RGBDR_ARENA_TRIALS", "library default (16)")
    load_package()
    from rgbd_recon_amd import capi, synth
    from rgbd_recon_amd import dist as rdist
    rig.capi, rig.synth, rig.rdist = capi, synth, rdist

    W, H = rig.W, rig.H = 512, 424
    rig.slab_rank, rig.slab_count = rig.slab if rig.loop else (rank, world)
    N, grid, rig.baseline_config, rig.scaling = choose_workload(rig.slab_count, False, args.weak, args.sensors,
        args.grid)
    rig.N, rig.grid, rig.G = N, grid, grid[0]
    if shared.get("scene_n") != N:
        shared["scene"], shared["scene_n"] = synth.Scene(N, W, H, lut_res=(128, 106, 128), seed=1234), N
        shared["d_depth"] = torch.from_numpy(shared["scene"].depth).to(dev)
        shared["d_color"] = torch.from_numpy(shared["scene"].color).to(dev)
    rig.scene, rig.d_depth, rig.d_color = shared["scene"], shared["d_depth"], shared["d_color"]
    flags = capi.FLAGS_DEFAULT | (capi.FLAG_PIPELINE if args.pipeline else 0)
    cfg = capi.make_config(N, (W, H), voxel_size=2.0 / rig.G, brick_size=8 * 2.0 / rig.G, flags=flags,
                           res_override=grid, slab_rank=rig.slab_rank, slab_count=rig.slab_count)
    ctx = rig.ctx = capi.Context(cfg, rig.local_rank)
    rig.geo = ctx.geo
    for i in range(N):
        ctx.set_calibration(i, rig.scene.xyz[i], rig.scene.lut_res, rig.scene.uv[i], rig.scene.lut_res, (0.5, 4.5))
        ctx.synth_inverse_calibration(i, rig.scene.pinhole(i))
    torch.cuda.synchronize()


def torch_exchanger(rig, reset=False):
    """the halo exchange through torch.distributed's process group (dist.HaloExchanger)"""
    ctx = rig.ctx
    if reset:                                # every rank leaves the library-managed exchange together
        ctx.set_sensor_shard(0, 0)
        ctx.set_halo_staging(-1)
    ctx.set_stream(rig.main_stream.cuda_stream)
    return rig.rdist.HaloExchanger(ctx.device_tsdf(), rig.dev, rig.main_stream, rank=rig.slab_rank,
        world=rig.slab_count,
                                   group=rig.transport["group"], via_host=rig.transport["kind"] != "rccl", ctx=ctx,
                                   loopback=rig.loop)


def open_transport(rig):
    """The stream the exchange is ordered on and one probe of the device transport before anything is timed.  If RCCL
    point-to-point on these buffers fails on this node, say so in the JSON line and stop: a run whose halos go through
    host
    memory would measure PCIe, not xGMI.  (--backend gloo asks for the host-staged path explicitly, for debugging
    several
    ranks on one GPU.)"""
    args, torch, dist, ctx, shared = rig.args, rig.torch, rig.dist, rig.ctx, rig.shared
    rig.halo = rig.rdist.halo_views(ctx.device_tsdf(), rig.dev)
    # The library enqueues on a torch stream so that the RCCL exchange can be ordered against the kernels with events
    # instead of host syncs (dist.HaloExchanger: boundary layers are staged device-to-device, the transfer of step k
    # overlaps step k+1).
    rig.main_stream = torch.cuda.Stream(rig.dev)
    torch.cuda.set_stream(rig.main_stream)
    rig.transport = {"kind": "rccl" if args.backend == "nccl" else args.backend + " (host-staged)", "group": None}
    rig.managed = bool(args.managed and args.backend == "nccl")
    if args.backend == "nccl" and "fallback" not in shared:
        shared["fallback"] = dist.new_group(backend="gloo")
    # (the copy-engine halo does not depend on RCCL point-to-point)
    if args.backend == "nccl" and args.halo_transport != "peer":
        ok, why = True, ""
        try:
            ctx.sync()
            rig.rdist.exchange_halo(*rig.halo, rank=rig.slab_rank, world=rig.slab_count, loopback=rig.loop)
            torch.cuda.synchronize()
        except Exception as e:  # noqa: BLE001 -- reported, not swallowed
            ok, why = False, "%s: %s" % (type(e).__name__, str(e).splitlines()[0][:200])
        if not all_ranks_ok(rig, ok):
            sys.stderr.write("[bench rank %d] RCCL point-to-point failed (%s); refusing to time a host-staged "
                             "fallback\n" % (rig.rank, why or "on another rank"))
            if rig.rank == 0:
                emit({"error": "RCCL point-to-point halo exchange failed: %s" % (why or "on another rank"),
                    "n_gpus": rig.world})
            ctx.close()
            dist.destroy_process_group()
            sys.exit(3)
    if not rig.managed:
        rig.exchanger = torch_exchanger(rig)


def open_communicators(rig):
    """The pre_* chain sharded by sensor over the ranks: rank r runs it for N / k sensors, the packed frames are
    all-gathered
    and the brick counters all-reduced on the chain's stream (SURVEY 8e's alternative to the redundant chain; the
    chain's
    time then shrinks with the number of GPUs like the sweep's).  On one GPU standing in for a rank (--slab /
    --loopback)
    the other ranks' sensors come from two unsharded frames of the same static scene and the gather's traffic is
    reproduced by RCCL send / recv to this process itself (dist.FrameGather loopback).  With the library-managed
    exchange
    (the default over RCCL) the raw communicators are created here; one that does not come up on ANY rank sends every
    rank
    to torch.distributed's collectives."""
    args, ctx, rdist = rig.args, rig.ctx, rig.rdist
    want_shard = args.shard and rig.N % rig.slab_count == 0 and rig.N > 1
    if not (want_shard or rig.managed):
        return
    if rig.loop:
        # both frame buffers of the two-stream schedule hold every sensor's frame
        for _ in range(2):
            ctx.update_device(rig.d_depth.data_ptr(), rig.d_color.data_ptr())
            ctx.clear_occupied_bricks(); ctx.process_textures(); ctx.update_occupied_bricks(); ctx.integrate()
        ctx.sync()
    if rig.managed:
        made = None
        try:
            ctx.enable_timers(True)
            # test hook: tests/test_bench_gpu.py walks the fallback roads
            if os.environ.get("RGBDR_BENCH_FAIL_MANAGED") == "construct":
                raise RuntimeError("RGBDR_BENCH_FAIL_MANAGED=construct")
            made = rdist.ManagedSlabExchange(ctx, rig.dev, rig.slab_rank, rig.slab_count, group=rig.transport["group"],
                shard=want_shard,
                                             loopback=rig.loop)
        except Exception as e:  # noqa: BLE001 -- a raw communicator that does not come up must not cost the run
            sys.stderr.write("[bench rank %d] library-managed RCCL unavailable (%s: %s)\n" % (rig.rank,
                type(e).__name__, str(e)[:200]))
        if all_ranks_ok(rig, made is not None):
            rig.exchanger = made
            rig.gather = made.gather if made.shard else None
            rig.rccl_info = made.comm.describe()
        else:
            sys.stderr.write("[bench rank %d] using torch.distributed for the exchange\n" % rig.rank)
            if made is not None:
                try:
                    made.close()
                except Exception:  # noqa: BLE001
                    pass
            rig.managed = False
            rig.exchanger = torch_exchanger(rig, reset=True)
    if not rig.managed and want_shard:
        rig.gather = rdist.FrameGather(ctx, rig.dev, rank=rig.slab_rank, world=rig.slab_count,
            group=rig.transport["group"],
                                       via_host=rig.transport["kind"] != "rccl", loopback=rig.loop)


class CopyEngineHalo:
    """the exchanger of a --halo-transport peer run: the halo steps go to dist.PeerCopySlabExchange, everything else
    (the gather
    of the sharded chain, its communicator) to the exchanger the run had"""

    def __init__(self, peer, inner):
        self.peer, self.inner = peer, inner
        self.begin_step, self.exchange_async = peer.begin_step, peer.exchange_async
        self.wait, self.last_transfer_ms = peer.wait, peer.last_transfer_ms

    def __getattr__(self, name):
        return getattr(self.inner, name)

    def close(self):
        self.peer.close()
        if hasattr(self.inner, "close"):
            self.inner.close()


def use_copy_engine_halo(rig):
    # the exports are Python bytes: gloo carries them
    group = rig.shared.get("fallback") if rig.world > 1 else None
    peer = rig.rdist.PeerCopySlabExchange(rig.ctx, rig.dev, rig.slab_rank, rig.slab_count, group=group,
        loopback=rig.loop)
    rig.exchanger = CopyEngineHalo(peer, rig.exchanger)
    rig.halo_by = "peer"


def trial_step(rig):
    """The library-managed exchange has never run between two devices (the pool has one GPU per box): its first step is
    a
    trial.  If it fails on ANY rank, every rank goes back to torch.distributed for the exchange and to the redundant
    chain, and the line says so -- a scaling run must not be lost to it.  (A trial that HANGS ends this process through
    the watchdog, and the supervisors start the next rung with fresh processes.)"""
    ok = True
    try:
        rig.step(False)
        rig.exchanger.wait()
        rig.ctx.sync()
        if os.environ.get("RGBDR_BENCH_FAIL_MANAGED") == "trial":
            raise RuntimeError("RGBDR_BENCH_FAIL_MANAGED=trial")
    except Exception as e:  # noqa: BLE001
        ok = False
        sys.stderr.write("[bench rank %d] library-managed exchange failed in its trial step (%s: %s)\n" % (rig.rank,
            type(e).__name__, str(e)[:200]))
    if not all_ranks_ok(rig, ok):
        rig.managed = False
        try:
            rig.exchanger.close()
        except Exception:  # noqa: BLE001
            pass
        rig.gather = None
        rig.exchanger = torch_exchanger(rig, reset=True)


def settle(rig):
    """Untimed set-up: memory a previous process released is wiped by the driver in the background for a while (a 6 GB
    free
    slows the sweep by 4 % for ~0.2 s).  Wait until the sweep time has settled before the warm-up and the
    timed steps begin."""
    rig.step(False)
    rig.ctx.settle(3.0)
    if rig.multi:
        rig.dist.barrier()


def headline_line(rig, dt, stats):
    """the JSON object of the headline: the driver's fields, the roofline block of the integrate kernel, the per-rank
    rows"""
    args, ctx, g, torch, dist = rig.args, rig.ctx, rig.geo, rig.torch, rig.dist
    N, W, H, grid, world, loop, multi = rig.N, rig.W, rig.H, rig.grid, rig.world, rig.loop, rig.multi
    V_local = g.res_volume[0] * g.res_volume[1] * (g.slab_voxel_z1 - g.slab_voxel_z0)
    V_total = g.res_volume[0] * g.res_volume[1] * g.res_volume[2] if not loop else V_local   # loopback: this slab only
    rig.V_total, rig.V_local = V_total, V_local
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
    local_ms_per_step = rig.local_dt / args.steps * 1e3
    halo_ms = rig.exchanger.last_transfer_ms() if multi else None
    # every rank's numbers on rank 0: the N > 1 line carries per-rank arrays and prices the SLOWEST rank's kernel
    # (rank 0 is an edge slab with one neighbour and one staged face; inner slabs stage two)
    per_rank = None
    if world > 1:
        mine = torch.tensor([int_s * 1e3, -1.0 if halo_ms is None else halo_ms, local_ms_per_step, float(bytes_launch),
                             replay_ms], dtype=torch.float64, device=rig.dev if args.backend == "nccl" else "cpu")
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
    if multi and rig.rccl_info is None and rig.transport["kind"] == "rccl":
        rig.rccl_info = rig.rdist.torch_rccl_info()
    rig.int_s, rig.bytes_launch, rig.achieved = int_s, bytes_launch, achieved
    rig.halo_ms, rig.ms_per_step = halo_ms, ms_per_step
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
        # when this approaches ms_per_step the host loop is the limit
        "host_enqueue_ms_per_step": round(rig.host_enqueue_ms, 4),
        "headline_retimed": getattr(rig, "retimed", None),           # bench_chain.retime_after_a_host_stall
        # ... and when that happened, what the FIRST measurement read (a stall inside the collective library is a
        # property
        # of the N > 1 path: a product loop cannot time itself again)
        "value_first_measurement": (round(V_total / (rig.retimed["discarded"][0]["ms_per_step"] * 1e-3) / 1e6, 1)
                                    if getattr(rig, "retimed", None) else None),
        "ms_per_step_first_measurement": rig.retimed["discarded"][0]["ms_per_step"] if getattr(rig, "retimed",
        None) else None,
        "higher_is_better": True,
        "scaling": rig.scaling if world > 1 else None,      # one GPU: nothing scales
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "voxel_sensor_updates_per_s": round(V_total * N / (dt / args.steps), 1),
        "config": {"workload": "%d sensors 512x424 -> %dx%dx%d TSDF, full pre_* chain + full-sweep integrate, 1:1 "
                               "inverse LUT" % ((N,) + tuple(g.res_volume)),
                   "baseline_config": rig.baseline_config,
                   "grid": list(g.res_volume), "sensors": N, "tsdf_limit": 0.01,
                   "schedule": "pipelined (pre_* of step k+1 overlaps integrate of step "
                   "k)" if args.pipeline else "sequential",
                   "parallelism": ("zslab%d" % world if world > 1 else "single") + (
                       " (loopback: slab %d of %d on one GPU, its own neighbour over RCCL)" % (rig.slab_rank,
                       rig.slab_count) if loop else ""),
                   "halo_transport": (("copy engine (hipMemcpyAsync from the neighbours' IPC-mapped staging sets, "
                   "rgbdr_halo_pull_async)"
                                       if getattr(rig, "halo_by",
                                       "rccl") == "peer" else rig.transport["kind"]) if multi else None),
                   "pre_chain": ("sharded by sensor on a chain-only context one frame ahead of the sweep: %d of %d "
                   "sensors per rank, the "
                                 "gather of frame k+1 under the sweep of frame k "
                                 "(dist.LaggedChain)" % (N // rig.slab_count, N)) if rig.lag is not None
                   else ("sharded by sensor: %d of %d sensors per rank, packed frames all-gathered + brick counters "
                   "all-reduced on "
                         "the chain's stream" % (N // rig.slab_count,
                         N)) if rig.gather is not None else "every sensor on every rank",
                   "pre_chain_choice": getattr(rig, "chain_choice", None),
                   "collectives": (("library-managed RCCL (C ABI: rgbdr_halo_exchange_async, "
                   "rgbdr_shard_allgather)" if rig.managed
                                    else "torch.distributed") if multi else None),
                   # which RCCL carried the exchange: the file mapped in this process, its version, and the number of
                   # ranks
                   # the communicator itself reports (ncclCommCount) -- N, or 1 when one GPU stands in for a rank
                   "rccl": rig.rccl_info,
                   "rccl_ranks": rig.rccl_info.get("ranks") if rig.rccl_info else None},
        "roofline": {"bound": "hbm",
                     "kernel": "rgbdr::k_integrate_tiled<%d, 4, true, false, %s>" % (N, "true" if multi else "false"),
                     "achieved": round(achieved / 1e9, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK, 4), "traffic": None, "traffic_source": None,
                     "bytes_per_launch": bytes_launch, "avg_launch_ms": round(int_s * 1e3, 4),
                     "launches_timed": int_n,
                     "rank": per_rank["slowest_rank"] if per_rank else (rig.slab_rank if loop else 0),
                     "box_stream_GBps": round(box_stream / 1e9, 1), "box_stream_replay_ms": round(replay_ms, 4),
                     "frac_of_box_stream": round(achieved / box_stream, 4) if box_stream > 0 else None,
                     "arena_placement_probe_ms": ctx.arena_probe()[0], "arena_kept": ctx.arena_probe()[1],
                     "arena_trials_used": rig.trials},
        "passes_ms": {k: round(v[0] / max(v[1], 1) * 1e-6, 4) for k, v in stats.items()},
    }
    placement_keys(out, ctx, int_s, bytes_launch, rig.trials, world, args)
    if per_rank is not None:
        out["per_rank"] = per_rank
        # BASELINE.json's multi-GPU configs name 8 sensors, its single-GPU config 4: a voxel of the N > 1 runs costs
        # twice the LUT bytes of a voxel of the N = 1 run, so `value` (Mvoxels/s) is not comparable across that step
        out["scaling_note"] = (
            "weak scaling of the N = 1 workload: every GPU owns 512^3 voxels of a %dx%dx%d volume and sweeps them "
            "from 4 sensors, "
            "value(N) compares with N x value(1); BASELINE.json's multi-GPU configs (8 sensors) are under "
            "baseline_configs_run" % tuple(grid)
            if args.weak else
            "N = 1 runs configs[2] (4 sensors), N = 2 / 4 configs[3] (8 sensors, 512^3), N = 8 configs[4] (8 sensors, "
            "1024^3): compare "
            "voxel_sensor_updates_per_s across N, not value; the fixed-work-per-GPU twin is under "
            "weak_scaling_4_sensors")
    traffic_keys(out, N, rig.G, world, loop, g)
    return out


def run_rank(args, slab=None, quiet=False, shared=None):
    """One rank of the benchmark: the whole single-GPU run, rank `RANK` of a --gpus N run, or (slab = (r, k))
    slab r of k on this GPU with itself as its neighbours.  Returns the JSON object (printed by rank 0 unless quiet).
    Every phase runs under the watchdog; everything after the headline is an isolated leg (bench_legs.py)."""
    shared = shared if shared is not None else {}
    rig = Rig(args, slab, quiet, shared)
    wd = rig.watchdog
    shared.setdefault("top_rig", rig)
    with wd.phase("init", 300.0):
        open_context(rig)
    if rig.multi:
        with wd.phase("transport", 120.0):
            open_transport(rig)
        with wd.phase("communicators", 120.0):
            open_communicators(rig)
        if rig.managed:
            with wd.phase("trial step", 60.0):
                trial_step(rig)
    # The "once per process, ~230 frames in, the host is held for 36-100 ms" of rounds 4-5 was Python's cyclic
    # collector:
    # a generation-2 pass over everything the set-up left on the heap (scene arrays, ctypes objects, torch), triggered
    # by
    # the allocation count of the step loop -- outside any HIP or RCCL call (profiles/stall_trace.sh,
    # r06_notes/host_stall.md:
    # with the collector off or the heap frozen the gap is gone).  The set-up's objects are moved out of the collector's
    # reach here; RGBDR_BENCH_GC=on leaves it as it was.
    if os.environ.get("RGBDR_BENCH_GC", "freeze") != "on":
        import gc
        gc.collect()
        if os.environ.get("RGBDR_BENCH_GC") == "off":
            gc.disable()
        else:
            gc.freeze()
    if rig.multi and args.halo_transport == "peer":
        with wd.phase("transport", 120.0):
            use_copy_engine_halo(rig)
    with wd.phase("settle", 60.0):
        settle(rig)
    if rig.multi and rig.gather is not None:
        with wd.phase("chain choice", 120.0):
            choose_chain(rig)
    with wd.phase("headline", 60.0 + 0.05 * (args.steps + args.warmup)):
        dt, stats = rig.timed(False, args.steps, args.warmup)
        rig.stats = stats
    if rig.multi:
        with wd.phase("headline", 120.0 + 0.1 * (args.steps + args.warmup)):
            dt, stats = retime_after_a_host_stall(rig, dt, stats)
    if rig.lag is not None:
        with wd.phase("headline", 120.0 + 0.05 * (args.steps + args.warmup)):
            dt, stats = recheck_lagged_headline(rig, dt, stats)
    with wd.phase("after the headline", 120.0):
        out = headline_line(rig, dt, stats)
    rig.out = out
    with wd.phase("after the headline", 60.0):
        leave_lagged_chain(rig)
    if rig.supervised and rig.rank == 0 and not quiet:
        emit(dict(out, provisional=True))     # the supervisor keeps the LAST line: this one only if the legs never end

    # ---- everything below is extra keys: isolated legs
    # ----------------------------------------------------------------
    import bench_legs
    lean = bool(shared.get("lean")) or args.no_legs      # --slab-sweep / the twin run / --no-legs: only the headline
    bench_legs.run_all(rig, out, lean=lean)
    with wd.phase("teardown", 60.0):
        if rig.multi and hasattr(rig.exchanger, "close"):
            rig.ctx.sync()
            rig.exchanger.close()             # the raw RCCL communicators of the library-managed exchange
        rig.ctx.close()
        if rig.multi:
            rig.torch.cuda.set_stream(rig.torch.cuda.default_stream(rig.dev))
    # The same job at fixed work per GPU (extra key): BASELINE.json's metric names 4 sensors into 512^3 on one GPU, its
    # multi-GPU configs 8 sensors -- so next to configs[3] / configs[4] the run also times the weak-scaling grid with
    # the
    # N = 1 sensor count (134 M voxels and 4 sensors per GPU: 512x512x1024 / 512x1024x1024 / 1024^3), whose value is
    # directly comparable with N times the N = 1 value.
    twin = getattr(args, "twin", None)
    if rig.world > 1 and not rig.loop and twin and not lean:
        key = "weak_scaling_4_sensors" if twin == "weak" else "baseline_configs_run"
        try:
            with wd.phase("leg " + key, 240.0):
                a2 = argparse.Namespace(**vars(args))
                a2.weak, a2.twin = twin == "weak", None
                sub = dict(shared)
                sub.update(lean=True, keep_pg=True)
                w = run_rank(a2, quiet=True, shared=sub)
                wd.on_expire = rig.expired
                for k in ("scene", "scene_n", "d_depth", "d_color"):
                    shared.pop(k, None)
                out[key] = {"baseline_config": w["config"]["baseline_config"], "grid": w["config"]["grid"],
                    "sensors": w["config"]["sensors"],
                            "scaling": w["scaling"], "value": w["value"], "ms_per_step": w["ms_per_step"],
                            "frames_per_s": w["frames_per_s"],
                            "voxel_sensor_updates_per_s": w["voxel_sensor_updates_per_s"],
                            "per_rank": w.get("per_rank"), "roofline_frac_slowest_rank": w["roofline"]["frac"],
                            "pre_chain": w["config"]["pre_chain"], "collectives": w["config"]["collectives"],
                            "comparable_with": ("N x the value of the N = 1 run (same sensors, same voxels per "
                            "GPU)" if twin == "weak" else
                                                "the 1-GPU time of the same config (8 sensors: DESIGN.md 6 has the "
                                                "denominators); across "
                                                "N by voxel_sensor_updates_per_s, not by value")}
        except Exception as e:  # noqa: BLE001 -- an extra key must never cost the line
            out[key] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
    if rig.rank == 0 and not quiet:
        emit(out)
        rig.printed = True
    if rig.multi and not shared.get("keep_pg"):
        with wd.phase("teardown", 30.0):      # (the line is out: a hang from here on ends the process with status 0)
            rig.dist.destroy_process_group()
            shared["pg"] = False
    if shared.get("top_rig") is rig:
        shared.pop("top_rig")
    return out


def placement_keys(out, ctx, int_s, bytes_launch, trials, world, args):
    """What the first placement gives (RGBDR_ARENA_TRIALS=1: no probing) and what the library's default gives (unset:
    the
    best of up to 16 candidates): the stream replay of those candidates, priced like the kernel (which runs at
    frac_of_box_stream of its replay)."""
    probe_ms, kept = ctx.arena_probe()
    r = out["roofline"]
    chunks, chunk_ms = ctx.arena_chunks()
    # no candidate was fast: the arena is a range of the fastest physical chunks (rgbdr_get_arena_chunks)
    if chunks:
        r["arena_chunks"], r["arena_chunks_replay_ms"] = chunks, round(chunk_ms, 4)
        probe_ms = list(probe_ms) + [round(chunk_ms, 4)]
        kept = len(probe_ms) - 1
    if len(probe_ms) > 1 and probe_ms[0] > 0 and probe_ms[kept] > 0 and world == 1:
        first_ms = int_s * 1e3 * probe_ms[0] / probe_ms[kept]
        r["avg_launch_ms_first_placement"] = round(first_ms, 4)
        r["frac_first_placement"] = round(bytes_launch / (first_ms * 1e-3) / HBM_PEAK, 4)
        # (until round 4 bench.py probed more placements than the library's default and this key scaled the result back;
        # now the headline context runs on the library's default, so it is `frac` itself unless --arena-trials was
        # given)
        dflt_ms = int_s * 1e3 * min(m for m in probe_ms if m > 0) / probe_ms[kept]
        r["frac_library_default"] = round(bytes_launch / (dflt_ms * 1e-3) / HBM_PEAK, 4)
        first3_ms = int_s * 1e3 * min(m for m in probe_ms[:3] if m > 0) / probe_ms[kept]
        r["frac_first_3"] = round(bytes_launch / (first3_ms * 1e-3) / HBM_PEAK, 4)   # round 3's library default
        if args.arena_trials == 0 and "RGBDR_ARENA_TRIALS" not in os.environ:
            r["frac_best_of_16"] = r["frac"]   # the library's default IS up to 16 candidates now
        r["placement_note"] = ("RGBDR_ARENA_TRIALS = %s; %d placements were probed: `frac` is on the one the library "
                               "kept, frac_first_placement / frac_library_default scale the measured launch time by "
                               "replay(candidate 0) / replay(kept) and by replay(best of the candidates) / "
                               "replay(kept), "
                               "frac_first_3 by replay(best of the first three) / replay(kept)" % (trials,
                               len(probe_ms)))
    elif world == 1:
        r["frac_first_placement"] = r["frac"]       # a single placement was looked at
        r["frac_library_default"] = r["frac"]
        r["avg_launch_ms_first_placement"] = r["avg_launch_ms"]


def traffic_keys(out, N, G, world, loop, g):
    traffic_file = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(traffic_file):
        return
    try:
        t = json.load(open(traffic_file))
        key = "%dx%d" % (N, G)
        if key in t and world == 1 and not loop and tuple(g.res_volume) == (G, G, G):
            # NOT measured in this run: the PMC passes of profiles/collect_pmc.sh on the same kernel and workload
            out["roofline"]["traffic"] = t[key]["hbm_bytes_per_launch"]
            src = t[key].get("source", "profiles/collect_pmc.sh")
            out["roofline"]["traffic_source"] = ("profiles/traffic.json (rocprofv3 --pmc passes of %s, "
                                                 "not this run)" % src)
    except Exception:  # noqa: BLE001
        pass


def slab_sweep(args):
    """--slab-sweep k: every slab r of k of BASELINE configs[3] (k = 2, 4) / configs[4] (k = 8) on this one GPU, one
    after the other in this process, each exactly as rank r of a --gpus k run executes it (sweep with staging,
    exchange over RCCL to itself on the side stream).  One JSON line: a row per rank, the spread, and the frame rate
    a k-GPU run would show if nothing but the slowest rank's step bounded it.  A projection from single-GPU runs,
    not a scaling measurement."""
    k = args.slab_sweep
    parse_slab("0/%d" % k)
    args.gpus, args.twin = 1, None
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
           "projection": "single-GPU per-rank runs of %s -- no scaling curve was "
           "measured" % line["config"]["baseline_config"],
           "ranks": rows,
           "integrate_ms_max": max(ints), "integrate_ms_min": min(ints),
           "integrate_spread": round(max(ints) / min(ints), 4),
           "ms_per_step_max": max(steps), "ms_per_step_min": min(steps),
           "projected_value_if_bound_by_slowest_rank": round(V / (max(steps) * 1e-3) / 1e6, 1),
           "config": line["config"], "dtype": "f32", "data": "synthetic"}
    emit(out)
    return 0


if __name__ == "__main__":
    main()
