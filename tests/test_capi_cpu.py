"""Host logic of the C-ABI library without a GPU: it loads, exports every symbol
include/rgbdr.h declares, its geometry agrees with the oracle's restatement of
setVoxelSize / setBrickSize / divideBox, and it fails loudly (no CPU fallback)
when no HIP device exists."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "rgbdr.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rgbdr_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg.capi.lib()
    names = declared_symbols()
    assert len(names) >= 40
    for n in names:
        assert hasattr(lib, n), "librgbdr_hip.so does not export %s" % n
    # and the Python binding table covers exactly the header
    assert sorted(pkg.capi.SYMBOLS) == names


def test_the_binding_table_has_the_header_s_signatures(pkg):
    """every prototype of include/rgbdr.h against capi.SYMBOLS: argument count, and per argument and for the result whether it is
    a pointer, a 32-bit integer (signed / unsigned), a float or a 64-bit size -- a drifted ctypes signature passes garbage
    without any error"""
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "rgbdr.h")).read(), flags=re.S)
    protos = re.findall(r"([A-Za-z_][\w\s\*]*?)\b(rgbdr_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", text)
    scalars = {"int": "i32", "int32_t": "i32", "unsigned": "u32", "unsigned int": "u32", "uint32_t": "u32", "float": "f32",
               "size_t": "64", "uint64_t": "64", "double": "f64"}

    def kind_c(decl, named=True):
        decl = re.sub(r"\s+", " ", decl.strip())
        if decl in ("void", ""):
            return None
        if "*" in decl or "[" in decl:
            return "ptr"
        toks = decl.replace("const ", "").split(" ")
        return scalars[" ".join(toks[:-1]) if named and len(toks) > 1 else " ".join(toks)]

    def kind_py(t):
        if t is None:
            return None
        for types, k in (((C.c_int, C.c_int32), "i32"), ((C.c_uint32, C.c_uint), "u32"), ((C.c_float,), "f32"),
                         ((C.c_size_t, C.c_uint64), "64"), ((C.c_double,), "f64")):
            if t in types:
                return k
        assert t in (C.c_void_p, C.c_char_p) or issubclass(t, C._Pointer), t
        return "ptr"

    assert len(protos) == len(pkg.capi.SYMBOLS) >= 79
    for ret, name, args in protos:
        res, argtypes = pkg.capi.SYMBOLS[name]
        want = [kind_c(a) for a in (x.strip() for x in args.split(",")) if a and a != "void"]
        assert want == [kind_py(t) for t in argtypes], (name, want, argtypes)
        assert kind_c(ret, named=False) == kind_py(res), (name, ret, res)


def test_header_is_plain_c_and_the_python_structs_have_the_c_sizes(pkg, tmp_path):
    """include/rgbdr.h is the boundary a C / cgo / JNI host binds: it must compile as C99 without extensions, and the
    ctypes mirrors in capi.py must have the C compiler's struct sizes (a drifted field would shift everything behind it)"""
    import subprocess
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include "rgbdr.h"\nint main(void) { printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(rgbdr_config), '
                   'sizeof(rgbdr_geometry), sizeof(rgbdr_lut), sizeof(rgbdr_view), sizeof(rgbdr_tsdf_device_view), '
                   'sizeof(rgbdr_image_device_view), sizeof(rgbdr_shard_device_view), sizeof(rgbdr_calibration_device_view)); return 0; }\n')
    exe = tmp_path / "sizes"
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    sizes = [int(v) for v in subprocess.run([str(exe)], capture_output=True, text=True).stdout.split()]
    capi = pkg.capi
    mirrors = [capi.Config, capi.Geometry, capi.Lut, capi.View, capi.TsdfDeviceView, capi.ImageDeviceView, capi.ShardDeviceView, capi.CalibrationDeviceView]
    assert sizes == [C.sizeof(m) for m in mirrors], (sizes, [C.sizeof(m) for m in mirrors])
    # ... and every field sits where the C compiler puts it, under the same name (offsetof / sizeof of each member)
    cnames = ["rgbdr_config", "rgbdr_geometry", "rgbdr_lut", "rgbdr_view", "rgbdr_tsdf_device_view", "rgbdr_image_device_view",
              "rgbdr_shard_device_view", "rgbdr_calibration_device_view"]
    body, want = "", []
    for cls, cname in zip(mirrors, cnames):
        for field in cls._fields_:
            body += 'printf("%%zu %%zu\\n", offsetof(%s, %s), sizeof(((%s*)0)->%s));\n' % (cname, field[0], cname, field[0])
            d = getattr(cls, field[0])
            want.append((cname, field[0], d.offset, d.size))
    src2 = tmp_path / "offsets.c"
    src2.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "rgbdr.h"\nint main(void) {\n' + body + "return 0; }\n")
    exe2 = tmp_path / "offsets"
    r = subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src2), "-o", str(exe2)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = [tuple(int(v) for v in line.split()) for line in subprocess.run([str(exe2)], capture_output=True, text=True).stdout.splitlines()]
    assert len(got) == len(want) >= 70
    for (cname, fname, off, size), g in zip(want, got):
        assert (off, size) == g, (cname, fname, (off, size), g)


def test_config_struct_size_is_checked(pkg):
    capi = pkg.capi
    cfg = capi.make_config(1, (16, 16))
    cfg.struct_size = 4
    h = C.c_void_p()
    assert capi.lib().rgbdr_create(C.byref(cfg), 0, C.byref(h)) == capi.ERR_INVALID_ARGUMENT
    assert b"struct_size" in capi.lib().rgbdr_last_error(None)


@pytest.mark.parametrize("field,value", [("num_sensors", 0), ("num_sensors", 9), ("depth_w", 0), ("tsdf_limit", 0.0),
                                         ("voxel_size", -1.0), ("compress_rgb", 3), ("depth_h", 32769), ("color_w", 2 ** 31 - 1)])
def test_invalid_config_rejected(pkg, field, value):
    capi = pkg.capi
    cfg = capi.make_config(2, (16, 16))
    setattr(cfg, field, value)
    h = C.c_void_p()
    rc = capi.lib().rgbdr_create(C.byref(cfg), 0, C.byref(h))
    assert rc == capi.ERR_INVALID_ARGUMENT and not h.value
    assert capi.lib().rgbdr_status_string(rc) == b"invalid argument"


@pytest.mark.parametrize("voxel,res,why", [(1e-6, (0, 0, 0), b"32768"), (1e-12, (0, 0, 0), b"32768"), (float("inf"), (0, 0, 0), b"finite"),
                                           (0.01, (40000, 8, 8), b"32768"), (0.01, (32768, 32768, 1024), b"2^31"),
                                           (0.01, (32768, 32768, 32768), b"2^31")])
def test_grid_sizes_beyond_the_index_types_are_refused(pkg, voxel, res, why):
    """compute_geometry bounds what every later size is computed from: a quotient beyond INT_MAX is never converted, an axis
    holds at most 32768 voxels and a volume fewer than 2^31 tiles (tile indices are 31-bit) -- nothing wraps into a small
    allocation"""
    capi = pkg.capi
    cfg = capi.make_config(1, (16, 16), voxel_size=voxel, res_override=res)
    g = capi.Geometry()
    assert capi.lib().rgbdr_compute_geometry(C.byref(cfg), C.byref(g)) == capi.ERR_INVALID_ARGUMENT
    assert why in capi.lib().rgbdr_last_error(None)
    h = C.c_void_p()
    assert capi.lib().rgbdr_create(C.byref(cfg), 0, C.byref(h)) == capi.ERR_INVALID_ARGUMENT and not h.value
    cfg = capi.make_config(1, (16, 16), res_override=(32768, 8, 8))          # the bound itself is a valid grid
    assert capi.lib().rgbdr_compute_geometry(C.byref(cfg), C.byref(g)) == capi.OK and list(g.tiles) == [4096, 1, 1]


@pytest.mark.parametrize("field,value", [("bbox_min", (float("-inf"), 0.0, -1.0)), ("bbox_max", (1.0, float("inf"), 1.0)),
                                         ("bbox_min", (-3e38, 0.0, -1.0)), ("bbox_max", (1.0, float("nan"), 1.0)),
                                         ("voxel_size", float("inf")), ("brick_size", float("inf")), ("brick_size", 3e38),
                                         ("tsdf_limit", float("inf")), ("tsdf_limit", float("nan"))])
def test_non_finite_configurations_are_refused(pkg, field, value):
    """an infinite box side made divideBox's loop condition NaN (a brick grid of zero bricks was accepted), an infinite brick or
    truncation limit reached a float -> int conversion: everything has to be finite before anything is derived from it
    (tests/native/geometry_fuzz.cpp found these under UBSan)"""
    capi = pkg.capi
    cfg = capi.make_config(2, (16, 16), res_override=(64, 64, 64), slab_count=2, slab_rank=1)
    if isinstance(value, tuple):
        getattr(cfg, field)[:] = value
    else:
        setattr(cfg, field, value)
    h = C.c_void_p()
    assert capi.lib().rgbdr_create(C.byref(cfg), 0, C.byref(h)) == capi.ERR_INVALID_ARGUMENT and not h.value
    g = capi.Geometry()
    if field != "tsdf_limit" or value != value or value == float("inf"):
        assert capi.lib().rgbdr_compute_geometry(C.byref(cfg), C.byref(g)) == capi.ERR_INVALID_ARGUMENT


def test_host_geometry_is_clean_under_asan_and_ubsan(tmp_path):
    """csrc/geometry.cpp (grid, bricks, slabs, LOD atlas, camera position, frustum planes: the part of the product that runs
    on the host) with 3000 random and hostile configurations under AddressSanitizer + UBSan + float-cast-overflow"""
    import subprocess
    exe = str(tmp_path / "geometry_fuzz")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined,float-cast-overflow", "-fno-sanitize-recover=all",
                        "-fno-omit-frame-pointer", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-o", exe,
                        os.path.join(ROOT, "tests", "native", "geometry_fuzz.cpp"), os.path.join(ROOT, "rgbd-recon_amd", "csrc", "geometry.cpp")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe, "3000"], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    m = re.search(r"geometry fuzz: (\d+) configurations accepted, (\d+) refused, (\d+) brick tables built", r.stdout)
    assert m and int(m.group(1)) > 300 and int(m.group(2)) > 300 and int(m.group(3)) > 300, r.stdout


def test_host_parsers_are_clean_under_asan_and_ubsan(pkg, tmp_path):
    """the mirror's .ks / sensor yml / .stream readers (rgbdr_host.hpp, the data formats either side of the path) with 8000
    hostile inputs -- random tokens, negative / huge / non-numeric numbers, files cut off anywhere -- under AddressSanitizer +
    UBSan + float-cast-overflow: an exception is a fine answer, undefined behaviour is not (the reference's
    static_cast<unsigned>(float) of such a token is)"""
    import subprocess
    exe = str(tmp_path / "parser_fuzz")
    lib_dir = os.path.join(ROOT, "rgbd-recon_amd")
    r = subprocess.run(["g++", "-std=c++14", "-O1", "-g", "-fsanitize=address,undefined,float-cast-overflow", "-fno-sanitize-recover=all",
                        "-fno-omit-frame-pointer", "-o", exe, os.path.join(ROOT, "tests", "native", "parser_fuzz.cpp"),
                        "-L" + lib_dir, "-lrgbdr_hip", "-Wl,-rpath," + lib_dir, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    work = tmp_path / "inputs"
    work.mkdir()
    r = subprocess.run([exe, str(work)], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    m = re.search(r"parser fuzz: (\d+) inputs parsed, (\d+) refused", r.stdout)
    assert m and int(m.group(1)) > 4000, r.stdout


def test_status_codes_of_the_header_the_library_and_the_binding_agree(pkg):
    capi = pkg.capi
    text = open(os.path.join(ROOT, "include", "rgbdr.h")).read()
    codes = {n: int(v) for n, v in re.findall(r"^\s*RGBDR_(OK|ERR_[A-Z_]+) = (-?\d+)", text, flags=re.M)}
    assert len(codes) == 8 and sorted(codes.values()) == list(range(-7, 1))
    for name, value in codes.items():
        assert getattr(capi, name) == value, name
        assert capi.lib().rgbdr_status_string(value) != b"unknown status", name
    assert capi.lib().rgbdr_status_string(-8) == b"unknown status"
    assert capi.lib().rgbdr_status_string(capi.ERR_NO_MEMORY) == b"host memory exhausted"


def test_no_exception_can_cross_the_c_boundary():
    """include/rgbdr.h promises that nothing throws: every int-returning entry point with a body of its own is a
    function-try-block closed by RGBDR_CONTAIN (context.hpp: std::bad_alloc -> RGBDR_ERR_NO_MEMORY, others -> RGBDR_ERR_STATE
    with what() as the message); one-line forwarders call such a function or a helper that only assigns a message"""
    import glob
    guarded = 0
    for path in sorted(glob.glob(os.path.join(ROOT, "rgbd-recon_amd", "csrc", "api*.cpp"))):
        lines = open(path).read().split("\n")
        for i, line in enumerate(lines):
            if not re.match(r"^int rgbdr_\w+\(", line):
                continue
            if line.rstrip().endswith("}"):                           # a one-liner
                assert re.search(r"\{ return (set_flag|upload_common)\(", line), line
                continue
            j = i
            while not lines[j].rstrip().endswith(")"):
                j += 1
            assert lines[j + 1] == "try {", "%s:%d %s" % (path, i + 1, line)
            k = j + 1
            while lines[k] != "}":
                k += 1
            assert re.match(r"^RGBDR_CONTAIN\((ctx|nullptr)\)$", lines[k + 1]), "%s:%d %s" % (path, k + 2, lines[k + 1])
            guarded += 1
    assert guarded >= 60
    ctx_hpp = open(os.path.join(ROOT, "rgbd-recon_amd", "csrc", "context.hpp")).read()
    assert "catch (...) { return rgbdr::contain_exception(ctx); }" in ctx_hpp


def test_no_device_fails_loudly(pkg):
    import torch

    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is present")
    capi = pkg.capi
    with pytest.raises(capi.RgbdrError) as e:
        capi.Context(capi.make_config(1, (16, 16)))
    assert e.value.status == capi.ERR_NO_DEVICE
    assert "no CPU fallback" in str(e.value)


@pytest.mark.parametrize("bmax,voxel,brick", [((1, 2.2, 1), 0.01, 0.1), ((1, 2, 1), 2.0 / 64, 8 * 2.0 / 64),
                                              ((1, 2, 1), 2.0 / 512, 8 * 2.0 / 512), ((1, 2.2, 1), 0.007, 0.1),
                                              ((0.5, 1.3, 2.0), 0.013, 0.05)])
def test_geometry_matches_oracle(pkg, orc, bmax, voxel, brick):
    capi = pkg.capi
    bmin = (-1.0, 0.0, -1.0)
    cfg = capi.make_config(1, (16, 16), bbox_min=bmin, bbox_max=bmax, voxel_size=voxel, brick_size=brick)
    g = capi.compute_geometry(cfg)
    assert tuple(g.res_volume) == orc.volume_res(bmin, bmax, voxel)
    bs = orc.adjust_brick_size(brick, voxel)
    assert g.brick_size == np.float32(bs)
    assert g.brick_voxels == int(round(brick / voxel))
    assert tuple(g.res_bricks) == orc.divide_box(bmin, bmax, bs)       # m_res_bricks: the divideBox loops alone
    assert g.num_bricks == g.res_bricks[0] * g.res_bricks[1] * g.res_bricks[2]
    assert tuple(g.tiles) == tuple(-(-r // 8) for r in g.res_volume)
    assert (g.slab_tile_z0, g.slab_tile_z1) == (0, g.tiles[2])
    assert (g.slab_voxel_z0, g.slab_voxel_z1) == (0, g.res_volume[2])


def test_res_override(pkg):
    capi = pkg.capi
    cfg = capi.make_config(1, (16, 16), voxel_size=2.0 / 64, res_override=(64, 64, 128))
    assert tuple(capi.compute_geometry(cfg).res_volume) == (64, 64, 128)


@pytest.mark.parametrize("tiles,count", [(64, 1), (64, 2), (64, 8), (81, 2), (100, 8), (7, 7), (13, 4)])
def test_slab_ranges_partition_the_tile_layers(pkg, tiles, count):
    lib = pkg.capi.lib()
    prev, sizes = 0, []
    for r in range(count):
        a, b = C.c_int(), C.c_int()
        assert lib.rgbdr_slab_range(tiles, count, r, C.byref(a), C.byref(b)) == 0
        assert a.value == prev and b.value > a.value
        sizes.append(b.value - a.value)
        prev = b.value
    assert prev == tiles and max(sizes) - min(sizes) <= 1
    a, b = C.c_int(), C.c_int()
    assert lib.rgbdr_slab_range(tiles, count, count, C.byref(a), C.byref(b)) == pkg.capi.ERR_INVALID_ARGUMENT


def test_slab_geometry_and_too_many_slabs(pkg):
    capi = pkg.capi
    cfg = capi.make_config(1, (16, 16), voxel_size=2.0 / 100, slab_rank=1, slab_count=3)   # 100 voxels = 13 tile layers
    g = capi.compute_geometry(cfg)
    assert (g.slab_tile_z0, g.slab_tile_z1) == (5, 9)
    assert (g.slab_voxel_z0, g.slab_voxel_z1) == (40, 72)
    last = capi.compute_geometry(capi.make_config(1, (16, 16), voxel_size=2.0 / 100, slab_rank=2, slab_count=3))
    assert last.slab_voxel_z1 == 100          # clipped to the volume, not to the padded tile layer
    bad = capi.make_config(1, (16, 16), voxel_size=2.0 / 16, slab_rank=0, slab_count=3)    # 2 tile layers
    with pytest.raises(capi.RgbdrError):
        capi.compute_geometry(bad)


def test_camera_position_matches_oracle(pkg, orc):
    capi, synth = pkg.capi, pkg.synth
    for i in range(3):
        s = synth.Sensor(i, 3, 64, 53)
        xyz, _ = synth.forward_luts(s, (16, 13, 16))
        out = (C.c_float * 3)()
        lut = capi.make_lut(xyz, (16, 13, 16))
        assert capi.lib().rgbdr_camera_position(C.byref(lut), out) == 0
        assert np.array_equal(np.array(out[:], np.float32), orc.camera_pos(xyz))


def test_product_does_not_touch_the_oracle():
    """the product path must not import, link or load anything under oracle/"""
    pk = os.path.join(ROOT, "rgbd-recon_amd")
    for dp, _, files in os.walk(pk):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".cuh", ".h", "Makefile")):
                text = open(os.path.join(dp, f)).read()
                assert "pyoracle" not in text and "rgbdr_oracle" not in text and "load_oracle" not in text, f


def test_host_mirror_has_the_reference_class_shape(tmp_path):
    """kinect::Reconstruction is the abstract base the application holds its drawing modes by
    (framework/reconstruction/reconstruction.hpp:11-37; std::vector<std::shared_ptr<Reconstruction>> g_recons,
    source/kinect_client.cpp:99,251-255): the mirror keeps that shape.  Compile-time check, no GPU needed."""
    import subprocess
    src = tmp_path / "shape.cpp"
    src.write_text('''
#include <memory>
#include <type_traits>
#include <vector>
#include "rgbdr_host.hpp"
using namespace rgbdr::host;
static_assert(std::is_abstract<Reconstruction>::value, "draw() is pure virtual");
static_assert(std::has_virtual_destructor<Reconstruction>::value, "held by base pointer");
static_assert(std::is_base_of<Reconstruction, ReconIntegration>::value, "ReconIntegration : public Reconstruction");
static_assert(!std::is_abstract<ReconIntegration>::value, "ReconIntegration implements draw()");
static_assert(std::is_constructible<ReconIntegration, CalibrationFiles const&, CalibVolumes const*, BoundingBox const&, float, float>::value,
              "the reference's constructor signature (recon_integration.cpp:30)");
// the members kinect_client.cpp calls through the base pointer
void (Reconstruction::*p_draw)() = &Reconstruction::draw;
void (Reconstruction::*p_drawF)() = &Reconstruction::drawF;
void (Reconstruction::*p_reload)() = &Reconstruction::reload;
void (Reconstruction::*p_resize)(std::size_t, std::size_t) = &Reconstruction::resize;
void (Reconstruction::*p_mask)(unsigned) = &Reconstruction::setColorMaskMode;
void (Reconstruction::*p_off)(float, float) = &Reconstruction::setViewportOffset;
std::vector<std::shared_ptr<Reconstruction>> g_recons;
// a second drawing mode with the constructor triple of the reference's other modes (recon_trigrid.hpp:16) reaches the
// frame's images -- texture units 1-7 of NetKinectArray::setStartTextureUnit -- through the base class
struct OtherMode : Reconstruction {
  OtherMode(CalibrationFiles const& cfs, CalibVolumes const* cv, BoundingBox const& bbox) : Reconstruction(cfs, cv, bbox) {}
  void draw() override { rgbdr_image_device_view v = frameImage("quality", 0); (void)v; }
};
static_assert(std::is_constructible<OtherMode, CalibrationFiles const&, CalibVolumes const*, BoundingBox const&>::value, "");
rgbdr_image_device_view (NetKinectArray::*p_dev)(int, unsigned) const = &NetKinectArray::deviceImage;
rgbdr_image_device_view (NetKinectArray::*p_dev_name)(std::string const&, unsigned) const = &NetKinectArray::deviceImage;
std::vector<float> (NetKinectArray::*p_rb)(int, unsigned) const = &NetKinectArray::readbackImage;
std::vector<unsigned char> (NetKinectArray::*p_rbc)(unsigned) const = &NetKinectArray::readbackColor;
// CalibVolumes' accessors for the other modes (CalibVolumes.hpp:38-39; getXYZVolumeUnits / getUVVolumeUnits -> the volumes themselves)
std::array<uint32_t, 3> (CalibVolumes::*p_vres)() const = &CalibVolumes::getVolumeRes;
std::array<float, 2> (CalibVolumes::*p_dl)(unsigned) const = &CalibVolumes::getDepthLimits;
rgbdr_calibration_device_view (CalibVolumes::*p_vol)(unsigned) const = &CalibVolumes::deviceVolumes;
// TimerDatabase's public surface (timer_database.hpp:12-22; begin / end live inside the library, sample() folds a frame)
double (TimerDatabase::*p_dur)(std::string const&) const = &TimerDatabase::duration;
double (TimerDatabase::*p_mean)(std::string const&) const = &TimerDatabase::mean;
void (TimerDatabase::*p_add)(std::string const&) = &TimerDatabase::addTimer;
void (TimerDatabase::*p_wmean)(std::string const&) const = &TimerDatabase::writeMean;
void (TimerDatabase::*p_wmin)(std::string const&) const = &TimerDatabase::writeMin;
void (TimerDatabase::*p_wmax)(std::string const&) const = &TimerDatabase::writeMax;
int main()
{
  // the unit names of NetKinectArray.cpp:430-439 and the images they stand for
  if (imageOfTextureUnit("color") != RGBDR_IMG_COLOR || imageOfTextureUnit("depth") != RGBDR_IMG_DEPTH_B_RG ||
      imageOfTextureUnit("quality") != RGBDR_IMG_QUALITY || imageOfTextureUnit("normal") != RGBDR_IMG_NORMAL ||
      imageOfTextureUnit("silhouette") != RGBDR_IMG_SILHOUETTE || imageOfTextureUnit("morph_depth") != RGBDR_IMG_DEPTH_MORPH ||
      imageOfTextureUnit("color_lab") != RGBDR_IMG_LAB)
    return 2;
  return (p_draw && p_drawF && p_reload && p_resize && p_mask && p_off && p_dev && p_dev_name && p_rb && p_rbc && p_vres && p_dl && p_vol && p_dur && p_mean && p_add && p_wmean && p_wmin && p_wmax) ? 0 : 1;
}
''')
    r = subprocess.run(["g++", "-std=c++14", "-fsyntax-only", "-Wall", "-I", os.path.join(ROOT, "rgbd-recon_amd", "host"), str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # the unit-name table against the reference's own (read where it lies; absent on the GPU box)
    ref = "/root/reference/framework/NetKinectArray.cpp"
    if os.path.exists(ref):
        import re
        text = open(ref).read()
        names = re.findall(r'^\s*m_texture_unit_offsets\["(\w+)"\] = m_start_texture_unit', text, re.M)   # ("bg" is commented out)
        assert names == ["color", "depth", "quality", "normal", "silhouette", "morph_depth", "color_lab"], names
        host = open(os.path.join(ROOT, "rgbd-recon_amd", "host", "rgbdr_host.hpp")).read()
        for nme in names:
            assert 'name == "%s"' % nme in host
