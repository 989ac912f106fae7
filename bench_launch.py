"""bench_launch.py -- how `bench.py --gpus N` is started and kept from ending without a JSON line.

Nothing here imports torch or touches the GPU: launch_ranks() is the parent of a run started as a plain process,
supervise_rank() the per-rank supervisor that walks the ladder of fresh child processes (RUNGS), FileStore what the
supervisors of one node agree through, Watchdog the per-phase deadline thread of a GPU child (bench.py's ranks).
"""
import json
import os
import sys
import threading
import time

BENCH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench.py")

# The ladder of an N > 1 run: (name, extra flags of the child).  Rung 0 is the design; the rungs
# below it trade features for paths that have run on more hardware.  Every rung is a FRESH set of child processes.
RUNGS = (
    ("library-managed RCCL (C ABI) + pre_* chain sharded by sensor", []),
    ("torch.distributed collectives + pre_* chain sharded by sensor", ["--torch-collectives", "--no-lagged"]),
    ("torch.distributed collectives + every sensor's chain on every rank, weak-scaling run only",
     ["--torch-collectives", "--no-shard", "--no-lagged", "--weak"]),
    # no RCCL on the per-frame path at all: the faces move by copy engine between IPC-mapped staging sets, every rank
    # runs every sensor's chain (a node whose RCCL point-to-point does not come up still yields a number)
    ("copy-engine halo (HIP IPC peer copies) + every sensor's chain on every rank, weak-scaling run only",
     ["--halo-transport", "peer", "--torch-collectives", "--no-shard", "--no-lagged", "--weak"]),
)
# seconds per rung; sum + slack stays under --launch-timeout (1500) < the driver's 1800
RUNG_BUDGETS = (470.0, 330.0, 330.0, 300.0)
EXIT_WATCHDOG = 75                       # a child stopped by its own per-phase watchdog



def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def budget_scale():
    """every watchdog budget is multiplied by this (tests shorten them)"""
    try:
        return float(os.environ.get("RGBDR_BENCH_BUDGET_SCALE", "1"))
    except ValueError:
        return 1.0


# ---------------------------------------------------------------------------------------------------------------------
# the launcher and the per-rank supervisor: neither imports torch nor touches the GPU
# ---------------------------------------------------------------------------------------------------------------------
def stop_process(p, grace=5.0):
    """terminate, then kill, exactly the process we started"""
    import subprocess
    if p.poll() is not None:
        return
    p.terminate()
    try:
        p.wait(grace)
    except subprocess.TimeoutExpired:
        p.kill()
        p.wait()


def launch_ranks(n, argv, timeout=1500.0, child_cmd=None, poll_s=0.2):
    """Parent of a `--gpus n` run that was started as a plain process: one child per rank with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in its environment (what torch.distributed.run would set; the
    reference's frame loop is one process too, source/kinect_client.cpp:1013-1014).  Each child is this script again
    and becomes the SUPERVISOR of its rank (supervise_rank).  The parent never touches the GPU, so starting children
    is not an exec from a GPU process.  Rank 0's stdout is read here and passed on: if the children end (or are
    stopped after `timeout`) without a JSON line, the parent prints an {"error": ...} line itself, so the caller
    always gets exactly one line.  Returns 0 when every rank did; otherwise the first failing rank's code after
    stopping the rest (by their own PIDs), 124 after `timeout`."""
    import subprocess
    import uuid
    cmd = list(child_cmd) if child_cmd else [sys.executable, BENCH] + list(argv)
    env = dict(os.environ)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env.setdefault("MASTER_PORT", str(free_port()))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update({"WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n), "RGBDR_BENCH_JOB": uuid.uuid4().hex})
    procs, lines = [], []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen(cmd, env=e, stdout=subprocess.PIPE if r == 0 else sys.stderr,
            text=(r == 0) or None))

    def pump():
        for ln in procs[0].stdout:
            if ln.startswith("{"):
                lines.append(ln)
            sys.stdout.write(ln)
            sys.stdout.flush()

    reader = threading.Thread(target=pump, daemon=True)
    reader.start()
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
                sys.stderr.write("[bench launcher] rank %d exited with status %d; stopping the other ranks\n" % (r,
                    code))
                # (supervisors leave on their own within moments of each other; rank 0's prints the line on its way out)
                t_grace = time.monotonic() + (0.0 if child_cmd else 20.0)
                while time.monotonic() < t_grace and any(procs[q].poll() is None for q in live):
                    time.sleep(poll_s)
                break
        if live and rc == 0:
            if time.monotonic() > deadline:
                rc = 124
                sys.stderr.write("[bench launcher] %d rank(s) still running after %.0f s; stopping them\n" % (len(live),
                    timeout))
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
    reader.join(5.0)
    if not lines and child_cmd is None:
        sys.stdout.write(json.dumps({"error": "no rank produced a line (launcher status %d)" % rc, "n_gpus": n}) + "\n")
        sys.stdout.flush()
    return rc


class FileStore:
    """What the supervisors of one job on one node agree through: small JSON files in a directory of the temp dir,
    written
    by rename.  (torch.distributed.run's own store would do, but a supervisor must not import torch: it never touches
    the
    GPU and costs nothing.)  The directory name is unique per job: our launcher's job id, or the launching agent's pid +
    start time + rendezvous port."""

    def __init__(self):
        import tempfile
        job = os.environ.get("RGBDR_BENCH_JOB")
        if not job:
            ppid = os.getppid()
            try:
                start = open("/proc/%d/stat" % ppid).read().rsplit(")", 1)[1].split()[19]
            except (OSError, IndexError):
                start = "0"
            job = "%d_%s_%s" % (ppid, start, os.environ.get("MASTER_PORT", "0"))
        self.dir = os.path.join(tempfile.gettempdir(), "rgbdr_bench_" + job)
        os.makedirs(self.dir, exist_ok=True)

    def put(self, name, obj):
        path = os.path.join(self.dir, name)
        tmp = "%s.%d.tmp" % (path, os.getpid())
        os.makedirs(self.dir, exist_ok=True)
        with open(tmp, "w") as f:
            json.dump(obj, f)
        os.replace(tmp, path)

    def get(self, name):
        try:
            with open(os.path.join(self.dir, name)) as f:
                return json.load(f)
        except (OSError, ValueError):
            return None

    def wait(self, name, timeout, poll_s=0.05):
        t_end = time.monotonic() + timeout
        while True:
            v = self.get(name)
            if v is not None or time.monotonic() > t_end:
                return v
            time.sleep(poll_s)

    def cleanup(self):
        import shutil
        shutil.rmtree(self.dir, ignore_errors=True)


def headline_of(lines):
    """the last JSON line of a rank-0 child that carries a headline (`value`); a child prints a provisional line right
    after its timed region and the full line at its end"""
    for ln in reversed(lines):
        try:
            j = json.loads(ln)
        except ValueError:
            continue
        if isinstance(j, dict) and "value" in j and "error" not in j:
            return j
    return None


def supervise_rank(args, argv):
    """The supervisor of one rank of an N > 1 run (started by launch_ranks or by torch.distributed.run; it never touches
    the GPU, so it may start FRESH children as often as it likes).  It walks RUNGS: per rung every supervisor starts one
    child (this script, role "rank", a fresh rendezvous port chosen by rank 0's supervisor), and rank 0's supervisor
    decides
    the rung's verdict: ok as soon as its child has printed a line with a headline and ended (or the rung's budget ran
    out
    with the provisional line in hand), failed otherwise.  Children bound their own phases (Watchdog) and leave with
    os._exit; the supervisors bound the rung.  The verdict travels through a FileStore.  Rank 0's supervisor prints THE
    line:
    the child's, plus `launch` = which rung produced it and what failed before; or {"error": ..., "attempts": [...]} and
    a non-zero status when every rung failed.  Total time <= --launch-timeout."""
    import subprocess
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ["WORLD_SIZE"])
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    t_start = time.monotonic()
    t_end = t_start + args.launch_timeout - 20.0
    budgets = [float(b) for b in args.rung_budgets.split(",")] if args.rung_budgets else list(RUNG_BUDGETS)
    store = FileStore()
    stub = os.environ.get("RGBDR_BENCH_CHILD_CMD")          # tests: a stand-in for the GPU child
    base_cmd = json.loads(stub) if stub else [sys.executable, BENCH]
    attempts, final = [], None

    def log(msg):
        sys.stderr.write("[bench supervisor %d] %s\n" % (rank, msg))
        sys.stderr.flush()

    for k in range(max(0, args.first_rung), len(RUNGS)):
        name, flags = RUNGS[k]
        budget = min(budgets[min(k, len(budgets) - 1)], t_end - time.monotonic())
        if rank == 0:
            # rank 0's supervisor alone decides whether a rung starts (the others would round the same clock
            # differently)
            if budget < min(60.0, budgets[min(k, len(budgets) - 1)]):
                attempts.append({"rung": k, "name": name,
                    "outcome": "not started: %.0f s of the run's budget left" % max(budget, 0.0)})
                store.put("rung%d.port" % k, -1)
                break
            port = free_port()
            store.put("rung%d.port" % k, port)
        else:
            port = store.wait("rung%d.port" % k, timeout=max(budget, 30.0))
            if port is None:
                log("rung %d: rank 0's supervisor never announced a port" % k)
                return 1
            if port < 0:
                break
        env = dict(os.environ, RGBDR_BENCH_ROLE="rank", RGBDR_BENCH_RUNG=str(k), MASTER_ADDR="127.0.0.1",
            MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        for key in [e for e in env if e.startswith("TORCHELASTIC_") or e in ("TORCH_NCCL_ASYNC_ERROR_HANDLING",)]:
            # the children rendezvous among themselves, not through the agent
            env.pop(key)
        t0 = time.monotonic()
        child = subprocess.Popen(base_cmd + list(argv) + flags, env=env,
            stdout=subprocess.PIPE if rank == 0 else sys.stderr,
                                 text=True if rank == 0 else None)
        lines = []
        if rank == 0:
            def pump(c=child, into=lines):
                for ln in c.stdout:
                    if ln.startswith("{"):
                        into.append(ln)
            reader = threading.Thread(target=pump, daemon=True)
            reader.start()
        deadline = t0 + budget
        verdict, why, reported = None, "", False
        peer_failed_at = None
        while verdict is None:
            code = child.poll()
            now = time.monotonic()
            if rank != 0:
                if code is not None and not reported:
                    store.put("rung%d.rc.%d" % (k, rank), code)
                    reported = True
                verdict = store.get("rung%d.verdict" % k)
                if verdict is None and now > deadline + 45.0:
                    log("rung %d: no verdict from rank 0's supervisor %.0f s after the rung's budget; giving up" % (k,
                        now - deadline))
                    stop_process(child)
                    return 1
            else:
                if code is not None:
                    reader.join(5.0)
                    line = headline_of(lines)
                    verdict = "ok" if line else "failed"
                    why = "" if line else "rank 0's child ended with status %d and no headline" % code
                elif now > deadline:
                    line = headline_of(lines)
                    verdict = "ok" if line else "failed"
                    why = "the rung's budget of %.0f s ran out %s" % (budget,
                        "with the provisional line in hand" if line else "before a headline")
                else:
                    # another rank's child died without a headline on our side: the job cannot complete; a short grace
                    # (its own watchdog or the broken collective will usually end our child first), then stop
                    if peer_failed_at is None:
                        for r in range(1, world):
                            c = store.get("rung%d.rc.%d" % (k, r))
                            if c not in (None, 0):
                                peer_failed_at, why = now, "rank %d's child ended with status %d" % (r, c)
                                break
                    elif now > peer_failed_at + 20.0 and not headline_of(lines):
                        verdict = "failed"
                if verdict is not None:
                    store.put("rung%d.verdict" % k, verdict)
            if verdict is None:
                time.sleep(0.1)
        # the rung is decided: children still running get a moment to finish their teardown, then go
        grace_end = time.monotonic() + (20.0 if verdict == "ok" else 0.0)
        while child.poll() is None and time.monotonic() < grace_end:
            time.sleep(0.1)
        code = child.poll()
        stop_process(child)
        took = round(time.monotonic() - t0, 1)
        if rank == 0:
            reader.join(5.0)
            line = headline_of(lines)
            if verdict == "ok" and line:
                final = line
                final["launch"] = {"rung": k, "rung_name": name, "rung_flags": flags, "rung_s": took,
                    "child_status": code,
                                   "line": "provisional (the child did not reach its "
                                   "end)" if line.get("provisional") else "final",
                                   "note": why or None, "failed_attempts": attempts,
                                   "launched_by": "torch.distributed.run"
                                   if "RGBDR_BENCH_JOB" not in os.environ else "bench.py"}
                final.pop("provisional", None)
                break
            errs = []
            for ln in lines:
                try:
                    j = json.loads(ln)
                    if "error" in j:
                        errs.append(str(j["error"])[:300])
                except ValueError:
                    pass
            attempts.append({"rung": k, "name": name, "flags": flags, "seconds": took, "child_status": code,
                             "outcome": why or "failed", "child_errors": errs or None})
            log("rung %d (%s) failed after %.0f s: %s" % (k, name, took, why))
        elif verdict == "ok":
            store.wait("printed", timeout=90.0)      # (see below)
            store.put("bye.%d" % rank, 0)
            return 0
    if rank != 0:
        # Leave only after rank 0's supervisor has printed THE line: a launcher that sees one worker end with a non-zero
        # status (torch.distributed.run does) stops the others at once, and rank 0's may still be collecting its child.
        store.wait("printed", timeout=90.0)
        store.put("bye.%d" % rank, 1)
        return 1
    # THE line first, the housekeeping after it: the launcher may stop this process once the other supervisors have left
    if final is not None:
        sys.stdout.write(json.dumps(final) + "\n")
    else:
        sys.stdout.write(json.dumps({"error": "every rung of the launch ladder failed", "n_gpus": world,
            "attempts": attempts,
                                     "seconds": round(time.monotonic() - t_start, 1)}) + "\n")
    sys.stdout.flush()
    store.put("printed", 1)
    t_bye = time.monotonic() + 15.0           # the other supervisors have read the last verdict: the directory can go
    while time.monotonic() < t_bye and any(store.get("bye.%d" % r) is None for r in range(1, world)):
        time.sleep(0.05)
    store.cleanup()
    return 0 if final is not None else 1


class Watchdog:
    """Per-phase deadlines of a GPU process, enforced from a side thread: a phase that overruns its budget ends the
    process with os._exit (a hung collective or kernel cannot be interrupted any other way).  Before the headline exists
    the status is EXIT_WATCHDOG and the supervisor moves to the next rung; once it exists `on_expire` prints the line
    as far as it got and the status is 0."""

    def __init__(self, on_expire, tag=""):
        self.on_expire, self.tag = on_expire, tag
        self.lock = threading.Lock()
        self.name, self.deadline, self.budget = None, None, 0.0
        self.scale = budget_scale()
        t = threading.Thread(target=self._run, daemon=True)
        t.start()

    def _run(self):
        while True:
            time.sleep(0.25)
            with self.lock:
                name, deadline, budget = self.name, self.deadline, self.budget
            if deadline is not None and time.monotonic() > deadline:
                sys.stderr.write("[bench%s] watchdog: phase '%s' exceeded its %.0f s\n" % (self.tag, name, budget))
                sys.stderr.flush()
                try:
                    self.on_expire(name, budget)
                finally:
                    os._exit(EXIT_WATCHDOG)

    def phase(self, name, seconds):
        return _Phase(self, name, seconds * self.scale)


class _Phase:
    def __init__(self, wd, name, seconds):
        self.wd, self.name, self.seconds = wd, name, seconds

    def __enter__(self):
        with self.wd.lock:
            self.outer = (self.wd.name, self.wd.deadline, self.wd.budget)
            self.wd.name, self.wd.deadline, self.wd.budget = self.name, time.monotonic() + self.seconds, self.seconds
        hang = os.environ.get("RGBDR_BENCH_HANG", "")      # test hook "<rung>:<phase>": this phase never ends
        if hang and hang == "%s:%s" % (os.environ.get("RGBDR_BENCH_RUNG", "-"), self.name):
            time.sleep(1e6)
        return self

    def __exit__(self, *exc):
        with self.wd.lock:
            self.wd.name, self.wd.deadline, self.wd.budget = self.outer
        return False


