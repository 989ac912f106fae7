"""The TEXT of the reference's hot-path shaders, compiled (oracle/build_shader_ref.py: glsl/pre_*.fs, inc_*.glsl,
tsdf_integration.vs read where they lie under /root/reference, syntax-rewritten to C++ against the reference's vendored
glm, samplers and driver-defined built-ins bound to the oracle's conventions) against the oracle's restatement and
against the HIP path, through committed fixtures (tests/golden/shader_passes_*.npz: data only).

This does not pin the oracle by the grading rule -- the sampler is a stand-in, there is no GL here -- and DESIGN.md says
so; what it removes is the risk that a statement of a shader was mis-read when it was restated: every arithmetic
statement between two fetches ran from the reference's own text.

  CPU, build container:  oracle == compiled shader text, bit for bit; fixtures reproduce
  CPU, anywhere:         oracle == fixtures
  GPU:                   HIP path == fixtures"""
import os
import sys

import numpy as np
import pytest

from conftest import count_diff, same_bits

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "golden")]
import shader_cases  # noqa: E402

IMG = {"morph": 1, "depth_rg": 2, "lab": 3, "depth_b": 4, "sil": 5, "normal": 6, "quality": 7}


def fixture(name):
    return np.load(os.path.join(ROOT, "tests", "golden", "shader_passes_%s.npz" % name))


def shader_lib():
    import shader_ref
    if not shader_ref.available():
        # in the build container the library is part of build(): missing means the reference's text no longer compiles
        assert not os.path.isdir("/root/reference/glsl"), \
            "oracle/_ref/libref_shaders.so is missing although /root/reference is present: run `make -C oracle shaders`"
        pytest.skip("reference checkout absent: the compiled shader text exists in the build container only")
    return shader_ref


def oracle_frame(orc, pkg, name):
    scene, cfg, geo, inv, inv_res = shader_cases.build(pkg.synth, pkg.capi, name)
    flags, G = shader_cases.CASES[name][5], shader_cases.CASES[name][3]
    ref = orc.run_pipeline(scene, pkg.synth.BBOX_MIN, pkg.synth.BBOX_MAX, (G, G, G), inv, limit=cfg.tsdf_limit,
                           brick_size=geo.brick_size, bv=geo.brick_voxels, res_bricks=tuple(geo.res_bricks),
                           filter_textures=bool(flags & 1), processed=bool(flags & 2), refine=bool(flags & 4), use_bricks=False,
                           compress=name in shader_cases.COMPRESSED_DEPTH)
    return scene, inv, ref


def compare(got, want, n, what):
    for k in shader_cases.IMAGES:
        for i in range(n):
            assert same_bits(got[k][i], want[k][i]), "%s: %s sensor %d: %d texels differ" % (what, k, i, count_diff(got[k][i], want[k][i]))
    assert np.array_equal(got["counters"], want["counters"]), "%s: brick counters differ" % what
    assert same_bits(got["tsdf"], want["tsdf"]), "%s: %d voxels differ" % (what, count_diff(got["tsdf"], want["tsdf"]))


@pytest.mark.parametrize("name", sorted(shader_cases.CASES))
def test_oracle_equals_the_compiled_shader_text(orc, pkg, name):
    shader_ref = shader_lib()
    scene, inv, ref = oracle_frame(orc, pkg, name)
    n, _, _, G, _, flags, _ = shader_cases.CASES[name]
    cfg = pkg.capi.make_config(n, shader_cases.CASES[name][1], voxel_size=2.0 / G, brick_size=8 * 2.0 / G)
    geo = pkg.capi.compute_geometry(cfg)
    got = shader_ref.run_frame(scene, pkg.synth.BBOX_MIN, pkg.synth.BBOX_MAX, (G, G, G), inv, limit=cfg.tsdf_limit,
                               brick_size=geo.brick_size, res_bricks=tuple(geo.res_bricks), filter_textures=bool(flags & 1),
                               processed=bool(flags & 2), refine=bool(flags & 4), compress=name in shader_cases.COMPRESSED_DEPTH)
    assert got["bricks_out_of_range"] == 0 and got["offcentre_lookups"] == 0
    compare(got, ref, n, "compiled shader text vs oracle")
    # and the committed fixture is what the compiled text produces today
    fx = fixture(name)
    compare(got, {k: fx[k] for k in fx.files}, n, "compiled shader text vs committed fixture")


@pytest.mark.parametrize("name", sorted(shader_cases.CASES))
def test_oracle_equals_the_shader_fixtures(orc, pkg, name):
    scene, inv, ref = oracle_frame(orc, pkg, name)
    fx = fixture(name)
    assert bytes(fx["inputs_sha256"]).decode() == shader_cases.digest(scene, inv), "the synthetic scene drifted: regenerate the fixtures"
    compare(ref, {k: fx[k] for k in fx.files}, shader_cases.CASES[name][0], "oracle vs fixture")
    assert np.any(np.abs(fx["tsdf"]) < 0.01) and fx["counters"].sum() > 0          # the fixtures are not trivial


def test_the_rewrite_is_syntax_only():
    """what oracle/build_shader_ref.py changes in a shader: declarations, qualifiers, swizzle calls, literal suffixes --
    checked on a GLSL snippet of its own (no reference text involved)"""
    import build_shader_ref as b
    src = ("#version 130\n#extension GL_ARB_x : enable\nuniform sampler3D[5] cv;\nuniform float a;\nlayout(location = 0) out vec2 o;\n"
           "noperspective in vec2 tc;\nlayout (std140, binding = 2) uniform B {\n  vec3 lo;\n};\n"
           "float f(const in vec3 p, out float q) { q = 1.0; return p.xy.x * 2.5 + 1e-3 + 3 + x1.0; }\n"
           "void main(void) { o = texture(cv[0], vec3(tc, 0.5)).xy * a; // 1.0 stays in comments\n}\n")
    body, slots = b.transform(src)
    assert "#version" not in body and "#extension" not in body and "layout" not in body and "uniform" not in body
    assert "sampler3D cv[5];" in body and "float a;" in body and "vec2 o;" in body and "vec2 tc;" in body and "vec3 lo;" in body
    assert "float f(const vec3 p, float& q) { q = 1.0f; return p.xy().x * 2.5f + 1e-3 + 3 + x1.0; }" in body
    assert "void shader_main() { o = texture(cv[0], vec3(tc, 0.5f)).xy() * a; // 1.0 stays in comments" in body
    assert sorted(s[0] for s in slots) == ["a", "cv", "lo", "o", "tc"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(shader_cases.CASES))
def test_hip_path_equals_the_shader_fixtures(pkg, name):
    capi = pkg.capi
    scene, cfg, geo, inv, inv_res = shader_cases.build(pkg.synth, capi, name)
    n = shader_cases.CASES[name][0]
    fx = fixture(name)
    assert bytes(fx["inputs_sha256"]).decode() == shader_cases.digest(scene, inv)
    ctx = capi.Context(cfg, 0)
    for i in range(n):
        ctx.set_calibration(i, scene.xyz[i], scene.lut_res, scene.uv[i], scene.lut_res, (0.5, 4.5))
        ctx.set_inverse_calibration(i, inv[i], inv_res)
    ctx.set_use_bricks(False)
    ctx.step(scene.depth_u8 if name in shader_cases.COMPRESSED_DEPTH else scene.depth, scene.color)
    for k, which in IMG.items():
        for i in range(n):
            got = ctx.readback_image(which, i)
            assert same_bits(got, fx[k][i]), "%s sensor %d: %d texels differ" % (k, i, count_diff(got, fx[k][i]))
    assert np.array_equal(ctx.readback_brick_counters(), fx["counters"])
    got = ctx.readback_tsdf()
    assert same_bits(got, fx["tsdf"]), "%d voxels differ" % count_diff(got, fx["tsdf"])
    ctx.close()


# ---- consumers of the volume: tsdf_raymarch.fs + shading.glsl, framebuffer_transfer.fs / tsdf_inpaint.fs / tsdf_colorfill.fs ----
def view_fixture(name):
    return np.load(os.path.join(ROOT, "tests", "golden", "shader_views_%s.npz" % name))


def frame_inputs(pkg, name):
    """the frame the views are rendered from: the committed pass fixture (what the compiled pass shaders produced)"""
    scene, cfg, geo, inv, inv_res = shader_cases.build(pkg.synth, pkg.capi, name)
    fx = fixture(name)
    return scene, cfg, geo, inv, fx


def occupied_mask(orc, counters, min_voxels):
    ids, _ = orc.update_occupied(counters, min_voxels)
    mask = np.zeros(counters.shape, np.uint8)
    mask[ids] = 1
    return mask


@pytest.mark.parametrize("name", sorted(shader_cases.VIEW_CASES))
def test_oracle_ray_march_and_hole_filling_equal_the_compiled_shader_text(orc, pkg, name):
    """every view of the case: oracle == compiled tsdf_raymarch.fs (+ shading.glsl) == committed fixture, and for the
    filled ones oracle == compiled transfer / inpaint / colorfill == fixture.  (Depth peels for space skipping come
    from the oracle in both: bricks.{vs,gs,fs} draw instanced cubes through a rasteriser, which is not compiled.)"""
    scene, cfg, geo, inv, fx = frame_inputs(pkg, name)
    vf = view_fixture(name)
    n = scene.N
    have_lib = True
    try:
        shader_ref = shader_lib()
    except pytest.skip.Exception:
        have_lib = False
    mask = occupied_mask(orc, fx["counters"], cfg.min_voxels_per_brick)
    for key, eye, mode, skip, fill in shader_cases.VIEW_CASES[name]:
        view = shader_cases.make_view(pkg.capi, pkg.synth, eye, mode, skip)
        peels = orc.depth_peels(bytes(view), pkg.synth.BBOX_MIN, geo.brick_size, tuple(geo.res_bricks), fx["counters"], mask) if skip else None
        args = (bytes(view), fx["tsdf"], inv, scene.uv, [scene.color[i] for i in range(n)], list(fx["depth_b"]), list(fx["quality"]))
        oc, od, on = orc.raymarch(*args, limit=cfg.tsdf_limit, peels=peels)
        for what, got in (("color", oc), ("depth", od), ("samples", on)):
            assert same_bits(got, vf["%s_%s" % (key, what)]), "oracle vs fixture: %s %s: %d differ" % (key, what, count_diff(got, vf["%s_%s" % (key, what)]))
        assert 0.02 < (od < 1).mean() < 0.98
        if fill:
            fc, fd = orc.fill_colors(oc, od)
            assert same_bits(fc, vf[key + "_filled_color"]) and same_bits(fd, vf[key + "_filled_depth"])
        if have_lib:
            sc, sd, sn = shader_ref.raymarch(*args, limit=cfg.tsdf_limit, peels=peels)
            assert same_bits(sc, oc) and same_bits(sd, od) and same_bits(sn, on), "compiled shader text vs oracle: %s" % key
            if fill:
                gc, gd = shader_ref.fill_colors(sc, sd)
                assert same_bits(gc, fc) and same_bits(gd, fd), "compiled hole-filling text vs oracle: %s" % key
    if not have_lib:
        pytest.skip("oracle == fixtures checked; the compiled shader text exists in the build container only")


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(shader_cases.VIEW_CASES))
def test_hip_ray_march_and_hole_filling_equal_the_shader_fixtures(pkg, name):
    capi = pkg.capi
    scene, cfg, geo, inv, inv_res = shader_cases.build(pkg.synth, capi, name)
    vf = view_fixture(name)
    n = scene.N
    ctx = capi.Context(cfg, 0)
    for i in range(n):
        ctx.set_calibration(i, scene.xyz[i], scene.lut_res, scene.uv[i], scene.lut_res, (0.5, 4.5))
        ctx.set_inverse_calibration(i, inv[i], inv_res)
    ctx.set_use_bricks(False)
    ctx.step(scene.depth, scene.color)
    for key, eye, mode, skip, fill in shader_cases.VIEW_CASES[name]:
        view = shader_cases.make_view(capi, pkg.synth, eye, mode, skip)
        color, depth, ns = ctx.raymarch(view)
        for what, got in (("color", color), ("depth", depth), ("samples", ns)):
            assert same_bits(got, vf["%s_%s" % (key, what)]), "%s %s: %d differ" % (key, what, count_diff(got, vf["%s_%s" % (key, what)]))
        if fill:
            fc, fd = ctx.fill_colors(view.width, view.height)
            assert same_bits(fc, vf[key + "_filled_color"]) and same_bits(fd, vf[key + "_filled_depth"])
    ctx.close()


@pytest.mark.parametrize("seed", range(8))
def test_random_small_scenes_oracle_equals_the_compiled_shader_text(orc, pkg, seed):
    """beyond the committed cases: random sensor counts, image / LUT / grid sizes (mostly not powers of two), host
    toggles and limits -- build container only (needs the compiled shader text)"""
    shader_ref = shader_lib()
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(1, 4))
    wh = (int(rng.integers(24, 49)), int(rng.integers(20, 41)))
    lut_res = tuple(int(v) for v in rng.integers(8, 15, 3))
    G = int(rng.choice([16, 20, 24, 27, 32]))
    inv_res = None if rng.random() < 0.4 else tuple(int(v) for v in rng.integers(12, 40, 3))
    flags = int(rng.integers(0, 8))
    limit = float(rng.choice([0.01, 0.02, 0.035]))
    scene = pkg.synth.Scene(n, wh[0], wh[1], lut_res=lut_res, seed=int(rng.integers(1, 10000)))
    cfg = pkg.capi.make_config(n, wh, voxel_size=2.0 / G, brick_size=8 * 2.0 / G, tsdf_limit=limit)
    geo = pkg.capi.compute_geometry(cfg)
    res = tuple(geo.res_volume)
    inv = scene.inverse(inv_res or res)
    kw = dict(filter_textures=bool(flags & 1), processed=bool(flags & 2), refine=bool(flags & 4))
    ref = orc.run_pipeline(scene, pkg.synth.BBOX_MIN, pkg.synth.BBOX_MAX, res, inv, limit=cfg.tsdf_limit, brick_size=geo.brick_size,
                           bv=geo.brick_voxels, res_bricks=tuple(geo.res_bricks), use_bricks=False, **kw)
    got = shader_ref.run_frame(scene, pkg.synth.BBOX_MIN, pkg.synth.BBOX_MAX, res, inv, limit=cfg.tsdf_limit,
                               brick_size=geo.brick_size, res_bricks=tuple(geo.res_bricks), **kw)
    assert got["offcentre_lookups"] == 0
    if got["bricks_out_of_range"]:
        pytest.skip("a marked position left the brick grid: undefined in the shader (the oracle skips it)")
    compare(got, ref, n, "compiled shader text vs oracle (seed %d)" % seed)
