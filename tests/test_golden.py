"""Golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py from
the CPU oracle on a seeded synthetic scene -- the reference ships none): the
oracle must keep reproducing them (CPU), and the HIP path must match them (GPU)."""
import glob
import os

import numpy as np
import pytest

from conftest import count_diff, same_bits

HERE = os.path.dirname(os.path.abspath(__file__))
# (tests/golden/shader_*.npz are the fixtures of tests/test_shader_ref.py, gl_*.npz those of tests/test_gl_ref.py)
FILES = sorted(f for f in glob.glob(os.path.join(HERE, "golden", "*.npz")) if not os.path.basename(f).startswith(("shader_", "gl_")))
IMG = {"morph": 1, "depth_rg": 2, "lab": 3, "depth_b": 4, "sil": 5, "normal": 6, "quality": 7}
BMIN, BMAX = (-1.0, 0.0, -1.0), (1.0, 2.0, 1.0)
FLAGS = {"two_sensors_1to1": 15, "two_sensors_generic": 15, "three_sensors_nobricks": 7}


class Stored:
    def __init__(self, z):
        self.N = z["depth"].shape[0]
        self.depth, self.color = z["depth"], z["color"]
        self.xyz = [z["xyz%d" % i] for i in range(self.N)]
        self.uv = [z["uv%d" % i] for i in range(self.N)]
        self.inv = [z["inv%d" % i] for i in range(self.N)]


def test_fixtures_present():
    assert len(FILES) == 3


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[:-4] for f in FILES])
def test_oracle_reproduces_golden(orc, path):
    z = np.load(path)
    s = Stored(z)
    f = FLAGS[os.path.basename(path)[:-4]]
    G = z["tsdf"].shape[0]
    voxel = np.float32(2.0 / G)
    brick = orc.adjust_brick_size(float(8 * voxel), float(voxel))
    rb = orc.divide_box(BMIN, BMAX, brick)
    ref = orc.run_pipeline(s, BMIN, BMAX, (G, G, G), s.inv, limit=0.01, brick_size=brick, bv=8, res_bricks=rb,
                           filter_textures=bool(f & 1), processed=bool(f & 2), refine=bool(f & 4),
                           use_bricks=bool(f & 8))
    for k in IMG:
        for i in range(s.N):
            assert same_bits(ref[k][i], z["%s%d" % (k, i)]), (k, i)
    assert np.array_equal(ref["counters"], z["counters"])
    assert np.array_equal(ref["occupied"], z["occupied"])
    assert same_bits(ref["tsdf"], z["tsdf"])


@pytest.mark.gpu
@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[:-4] for f in FILES])
def test_hip_matches_golden(pkg, path):
    capi = pkg.capi
    z = np.load(path)
    s = Stored(z)
    G = z["tsdf"].shape[0]
    H, W = s.depth.shape[1:]
    cfg = capi.make_config(s.N, (W, H), voxel_size=2.0 / G, brick_size=8 * 2.0 / G,
                           flags=FLAGS[os.path.basename(path)[:-4]])
    ctx = capi.Context(cfg, 0)
    for i in range(s.N):
        rz, ry, rx = s.xyz[i].shape[:3]
        ctx.set_calibration(i, s.xyz[i], (rx, ry, rz), s.uv[i], (rx, ry, rz), (0.5, 4.5))
        iz, iy, ix = s.inv[i].shape[:3]
        ctx.set_inverse_calibration(i, s.inv[i], (ix, iy, iz))
    ctx.step(s.depth, s.color)
    for k, which in IMG.items():
        for i in range(s.N):
            got = ctx.readback_image(which, i)
            assert same_bits(got, z["%s%d" % (k, i)]), "%s %d: %d differ" % (k, i, count_diff(got, z["%s%d" % (k, i)]))
    assert np.array_equal(ctx.readback_brick_counters(), z["counters"])
    assert np.array_equal(ctx.get_occupied()[0], z["occupied"])
    got = ctx.readback_tsdf()
    assert same_bits(got, z["tsdf"]), "%d voxels differ" % count_diff(got, z["tsdf"])
    ctx.close()
