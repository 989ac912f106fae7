"""Pins the oracle against code compiled from the reference's own sources
(oracle/_ref, built by oracle/Makefile from /root/reference): calibration-volume
file format, record layouts, record order and trilinear interpolation."""
import ctypes as C
import os

import numpy as np
import pytest


@pytest.fixture(scope="module")
def ref(orc):
    l = orc.ref_lib()
    if l is None:
        pytest.skip("oracle/_ref not built (reference checkout absent)")
    return l


def test_record_sizes(ref):
    # kinect::xyz 12 B, kinect::uv 8 B (framework/DataTypes.h:12-35), glm::fvec4 16 B
    assert ref.ref_sizeof_xyz() == 12
    assert ref.ref_sizeof_uv() == 8
    assert ref.ref_sizeof_fvec4() == 16


@pytest.mark.parametrize("floats", [3, 2, 4])
def test_file_format_oracle_writes_reference_reads(ref, orc, tmp_path, floats):
    rng = np.random.default_rng(floats)
    data = rng.standard_normal((5, 4, 3, floats)).astype(np.float32)  # [rz, ry, rx, c]
    path = str(tmp_path / "a.cv")
    assert orc.lut_write(path, data, floats) == 0
    assert os.path.getsize(path) == 20 + data.size * 4  # 3*u32 + 2*f32 header
    res = (C.c_uint32 * 3)()
    lim = (C.c_float * 2)()
    out = np.zeros_like(data)
    assert ref.ref_lut_read(path.encode(), res, lim, out.ctypes.data_as(C.c_void_p), floats) == 0
    assert tuple(res) == (3, 4, 5)
    assert tuple(lim) == (0.5, 4.5)
    assert np.array_equal(out, data)


@pytest.mark.parametrize("floats", [3, 2, 4])
def test_file_format_reference_writes_oracle_reads(ref, orc, tmp_path, floats):
    rng = np.random.default_rng(10 + floats)
    data = rng.standard_normal((3, 6, 2, floats)).astype(np.float32)
    path = str(tmp_path / "b.cv")
    res = (C.c_uint32 * 3)(2, 6, 3)
    lim = (C.c_float * 2)(0.5, 4.5)
    assert ref.ref_lut_write(path.encode(), res, lim, data.ctypes.data_as(C.c_void_p), floats) == 0
    got, limits = orc.lut_read(path, floats)
    assert limits == (0.5, 4.5)
    assert np.array_equal(got, data)


def test_record_order_x_fastest(ref, orc, tmp_path):
    data = np.arange(4 * 3 * 2 * 3, dtype=np.float32).reshape(4, 3, 2, 3)
    path = str(tmp_path / "c.cv_xyz")
    orc.lut_write(path, data, 3)
    out = (C.c_float * 3)()
    for (x, y, z) in [(0, 0, 0), (1, 0, 0), (0, 2, 0), (1, 2, 3), (0, 1, 2)]:
        ref.ref_lut_at_xyz(path.encode(), x, y, z, out)
        assert list(out) == list(data[z, y, x])


def test_trilinear_matches_reference_cpu_helper(ref, orc):
    # kinect::getTrilinear works on un-normalised texel coordinates with weights
    # (1-w)*a + w*b; the oracle uses a + w*(b-a): equal to a few ulp.
    rng = np.random.default_rng(5)
    vol = rng.uniform(-2, 2, (6, 7, 8, 3)).astype(np.float32)  # rz, ry, rx
    ref.ref_get_trilinear.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_uint, C.c_float, C.c_float, C.c_float,
                                      C.c_void_p]
    out = (C.c_float * 3)()
    for _ in range(200):
        x, y, z = rng.uniform(0, 7), rng.uniform(0, 6), rng.uniform(0, 5)
        ref.ref_get_trilinear(vol.ctypes.data_as(C.c_void_p), 8, 7, 6, x, y, z, out)
        # texel coordinate t <-> normalised s = (t + 0.5) / n
        got = orc.tex3d(vol, (x + 0.5) / 8, (y + 0.5) / 7, (z + 0.5) / 6)
        np.testing.assert_allclose(got, np.array(out[:]), rtol=0, atol=2e-5)


@pytest.mark.parametrize("mode", [1, 5])
@pytest.mark.parametrize("wh", [(64, 52), (61, 54), (4, 4), (7, 3)])
def test_dxt_decode_matches_reference_squish(ref, orc, mode, wh):
    """DXT colour frames: the oracle's decoder against squish::DecompressImage, the
    decoder the reference itself applies to its DXT1 frames (NetKinectArray.cpp:633)"""
    W, H = wh
    nb = ((W + 3) // 4) * ((H + 3) // 4)
    rng = np.random.default_rng(W * 100 + H + mode)
    blocks = rng.integers(0, 256, nb * (8 if mode == 1 else 16), dtype=np.uint8)
    got = orc.decode_dxt(blocks, W, H, mode)
    rgba = np.zeros((H, W, 4), np.uint8)
    ref.ref_squish_decompress(rgba.ctypes.data_as(C.c_void_p), W, H, blocks.ctypes.data_as(C.c_void_p), mode)
    assert np.array_equal(got, rgba[..., :3])
