import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from __graft_entry__ import load_oracle, load_package  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Where the reference checkout exists (the build container) the libraries compiled from it / for running it are part
    # of build(); a fresh clone that runs the tests first gets them built here, once.  A failing build leaves them
    # missing and the tests that need them FAIL (they only skip where the checkout is absent, i.e. on the GPU box).
    ref_dir = os.path.join(ROOT, "oracle", "_ref")
    if os.path.isdir("/root/reference/glsl") and not all(
            os.path.exists(os.path.join(ref_dir, f)) for f in ("libref_shim.so", "libref_shaders.so", "libglctx.so")):
        import subprocess
        subprocess.call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "portable", "ref", "shaders", "glctx"])


@pytest.fixture(scope="session")
def pkg():
    load_package()
    from rgbd_recon_amd import capi, synth

    class P:
        pass

    p = P()
    p.capi, p.synth = capi, synth
    return p


@pytest.fixture(scope="session")
def orc():
    return load_oracle()


def canonical_bits(a):
    """the uint32 pattern of every float, with every NaN mapped to one pattern (payloads and signs of NaNs are not part of
    the contract: the oracle's libm and the device produce different quiet NaNs for the same invalid operation)"""
    a = np.ascontiguousarray(a, dtype=np.float32)
    bits = a.view(np.uint32).copy()
    bits[np.isnan(a)] = 0x7FC00000
    return bits


def same_bits(a, b):
    """BIT-exact float comparison: the uint32 patterns are equal, -0 differs from +0; only NaNs are canonicalised"""
    a, b = np.asarray(a, dtype=np.float32), np.asarray(b, dtype=np.float32)
    return a.shape == b.shape and bool(np.array_equal(canonical_bits(a), canonical_bits(b)))


def same_values(a, b):
    """value-exact float comparison: NaN matches NaN, -0 matches +0 (what rounds 1-4 called same_bits)"""
    a, b = np.asarray(a, dtype=np.float32), np.asarray(b, dtype=np.float32)
    return a.shape == b.shape and bool(np.all((a == b) | (np.isnan(a) & np.isnan(b))))


def count_diff(a, b):
    a, b = np.asarray(a, dtype=np.float32), np.asarray(b, dtype=np.float32)
    return int(np.sum(canonical_bits(a) != canonical_bits(b)))
