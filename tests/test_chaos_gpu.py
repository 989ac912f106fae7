"""Misuse resilience of the C ABI on the device: random calls in any order with valid, out-of-range and null arguments.
Every call has to come back with a status from the header's list -- no crash, no hang, no exception -- and whatever
state the chaos left behind, a frame uploaded and processed afterwards equals the oracle's.  Each sequence runs in a
process of its own (tests/chaos_worker.py) under a time limit: a call that hangs or kills the process is named by the
trace the worker writes before every call."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("seed,slab", [(1, None), (2, None), (3, (1, 3)), (4, (0, 2))] +
                         [(5000 + k, None if k % 3 else (k % 2, 2)) for k in range(int(os.environ.get("RGBDR_EXTRA_SEEDS", "0")) // 8)])
def test_random_misuse_never_crashes_and_leaves_a_usable_context(seed, slab, tmp_path):
    trace = str(tmp_path / "trace.txt")
    cmd = [sys.executable, os.path.join(HERE, "chaos_worker.py"), str(seed), str(slab[0] if slab else -1), str(slab[1] if slab else 1), trace]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=150)
    except subprocess.TimeoutExpired:
        last = open(trace).read().splitlines()[-1:] if os.path.exists(trace) else []
        pytest.fail("chaos sequence %d hangs; last call (step, index into `calls`): %s" % (seed, last))
    last = open(trace).read().splitlines()[-1:] if os.path.exists(trace) else []
    assert r.returncode == 0 and "chaos ok" in r.stdout, "seed %d, last call %s\n%s" % (seed, last, (r.stdout + r.stderr)[-3000:])
