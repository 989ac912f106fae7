"""The reference's GLSL RUN -- the files under /root/reference/glsl compiled by Mesa's GLSL compiler and executed by
llvmpipe in the build container (oracle/gl_ref.py, oracle/gl_context.c) -- against the oracle and the HIP path, through
committed fixtures (tests/golden/gl_passes_*.npz, gl_views_*.npz: data only, made by tests/golden/make_gl_golden.py).

This is what pins the oracle's pass arithmetic: the expected values are outputs of the reference's own shader code on a
real OpenGL implementation (texture units, rasteriser, image stores, atomics included), not of a restatement.

  CPU, build container:  Mesa runs the shaders again and reproduces the committed fixtures bit for bit
  CPU, anywhere:         oracle  vs fixtures, within the tolerances below
  GPU:                   HIP path vs fixtures, same tolerances (the HIP path is bit-identical to the oracle)

Tolerances (absolute; what is observed is in DESIGN.md section 2, the same statement).  They are llvmpipe-against-IEEE
differences, not slack for the algorithm: llvmpipe evaluates pow / exp / inversesqrt / normalize with its own polynomial
and Newton approximations and filters 8-bit textures with 8-bit weights, the oracle uses libm and float weights.
  morph, depth_rg (float depth frames), depth_b, silhouette     bit-exact, every scene
  morph, depth_rg of u8 depth frames                            1e-6   (unorm8 -> float: x * (1/255) vs x / 255)
  Lab colour                                                    4e-3   (pow(x, 1/3), 8-bit bilinear weights)
  normals                                                       5e-5   (1.2e-5 seen at 512 x 424, where the differenced positions are closest);
                                                                0.1 % of the components up to 1e-3 (near-degenerate cross products)
  quality                                                       2e-6 + 2e-5 relative; excluded: texels where llvmpipe's
                                                                pow(angle < 0, 2) is NaN (undefined in GLSL; include/rgbdr.h at
                                                                rgbdr_process_textures) or a NaN normal is a bilinear neighbour
                                                                (weight 0) -- each excluded texel is checked to have a negative angle
  brick counters                                                equal on the small scenes; at 512 x 424 the sum of |differences| is
                                                                within 1e-3 of the counter sum (124 of 388 134 seen: an increment
                                                                lands in the neighbouring brick when the world position's last bit
                                                                differs at a brick face); the occupied list is equal everywhere
  TSDF                                                          5e-7 (2.3e-7 seen with four 512 x 424 sensors, <= 2e-9 on the small
                                                                scenes), except voxels all of whose in-band sensors look at the
                                                                surface at a grazing angle (total weight < 1e-8: each is verified to
                                                                lie between its sensors' signed distances; at most 2e-4 of a sample --
                                                                3 of 59 845 seen on the 512^3 bands, max 6.9e-6); voxels fed by a
                                                                NaN-on-llvmpipe quality texel excluded
  TSDF class (-limit / surface band / +limit)                   the same everywhere, except a voxel whose two values lie within 1e-6 of
                                                                the same boundary (class_flips: 1 of 59 419 seen in the DXT1 sample)"""
import os
import sys

import numpy as np
import pytest

from conftest import count_diff, same_bits

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "golden")]
import shader_cases  # noqa: E402

IMG = {"morph": 1, "depth_rg": 2, "lab": 3, "depth_b": 4, "sil": 5, "normal": 6, "quality": 7}
EXACT = ("morph", "depth_rg", "depth_b", "sil")
TOL = {"lab": 4e-3, "normal": 5e-5, "quality": 2e-6}
LOOSE = {"normal": (1e-3, 1e-3)}  # a normal is normalize(cross(a, b)) of differenced positions: ill-conditioned where a x b is tiny
RTOL = {"quality": 2e-5}         # two pow(x, 6.0) and a pow(x, 2.0) of llvmpipe's exp2(y * log2(x)) in one product
TOL_U8_DEPTH = 1e-6
TOL_TSDF = 5e-7          # 5e-5 of the band's half-width: the weighted mean inherits the relative error of llvmpipe's quality (pow)
MAX_NAN_FRACTION = 5e-3          # texels whose quality is NaN on llvmpipe only; each one must have a negative angle


def fixture(name):
    return np.load(os.path.join(ROOT, "tests", "golden", "gl_passes_%s.npz" % name))


def gl_lib():
    import gl_ref
    if not gl_ref.available():
        # in the build container the context library is part of build(): missing means the pin silently vanished
        mesa = os.path.exists("/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so") and os.path.exists("/usr/include/GL/internal/dri_interface.h")
        assert not (os.path.isdir("/root/reference/glsl") and mesa), \
            "oracle/_ref/libglctx.so is missing although /root/reference and Mesa are present: run `make -C oracle glctx`"
        pytest.skip("reference checkout (or Mesa's software rasteriser) absent: the shaders run in the build container only")
    return gl_ref


def within(got, want, tol, what, allow_nan_in_want=False, excuse=None, rtol=0.0, loose=None):
    """`loose` = (fraction, tolerance): that share of the values may exceed `tol`, up to the second tolerance"""
    got, want = np.asarray(got, np.float32), np.asarray(want, np.float32)
    assert got.shape == want.shape, what
    skip = np.isnan(want) & ~np.isnan(got) if allow_nan_in_want else np.zeros(got.shape, bool)
    # (a sanity cap; every such value is excused one by one below.  Images of a thousand texels may hold a handful)
    assert skip.sum() <= max(MAX_NAN_FRACTION * skip.size, 8 if excuse is not None else 0), "%s: %d values are NaN on llvmpipe only" % (what, int(skip.sum()))
    if excuse is not None:
        for idx in np.argwhere(skip):
            assert excuse(tuple(int(v) for v in idx)), "%s: NaN on llvmpipe at %s without a negative angle" % (what, tuple(idx))
    same_nan = np.isnan(got) == np.isnan(want)
    assert np.all(same_nan | skip), "%s: %d NaN mismatches" % (what, int((~(same_nan | skip)).sum()))
    fin = np.isfinite(got) & np.isfinite(want)
    inf_ok = (got == want) | ~(np.isinf(got) | np.isinf(want))
    assert np.all(inf_ok | skip), "%s: infinities differ" % what
    d = np.abs(got[fin].astype(np.float64) - want[fin].astype(np.float64)) - rtol * np.abs(want[fin].astype(np.float64))
    if loose is not None and d.size and d.max() > tol:
        # (a share of the values -- or, for images of a thousand texels, half a dozen of them)
        assert (d > tol).sum() <= max(loose[0] * d.size, 6) and d.max() <= loose[1], "%s: %d values beyond %.3g, max %.3g" % (
            what, int((d > tol).sum()), tol, d.max())
        return int(skip.sum())
    assert d.size == 0 or d.max() <= tol, "%s: max |difference| %.3g > %.3g (%d values beyond)" % (what, d.max(), tol, int((d > tol).sum()))
    return int(skip.sum())


def negative_angle(scene, i, depth_b, normal):
    """The two situations in which llvmpipe's quality is NaN and the oracle's is not, checked per texel:
    (a) pre_quality.fs:43-48 in float64: dot(normalize(camera - world_pos), normal) < 0, so pow(angle, 2.0) (:104) is
        undefined in GLSL -- llvmpipe returns NaN, libm's powf the square;
    (b) a NaN normal among the 8 neighbours: `kinect_normals` is LINEAR and sampled at the texel centre, llvmpipe
        evaluates a + w * (b - a) with w = 0 (or 1e-7 for sizes that are not powers of two) in float, which is NaN for
        b = NaN; the oracle reads the texel itself (a texture unit's 8-bit weight is 0 there)."""
    import pyoracle
    H, W = depth_b.shape[:2]
    cam = np.asarray(pyoracle.camera_pos(scene.xyz[i]), np.float64)

    def check(idx):
        y, x = idx
        if np.isnan(normal[max(y - 1, 0):y + 2, max(x - 1, 0):x + 2]).any():
            return True
        pos = np.asarray(pyoracle.tex3d(scene.xyz[i], (x + 0.5) / W, (y + 0.5) / H, float(depth_b[y, x, 0])), np.float64)[:3]
        d = cam - pos
        return float(np.dot(d / np.linalg.norm(d), np.asarray(normal[y, x], np.float64))) < 1e-6
    return check


def degenerate_normal(scene, i, depth_b):
    """pre_normal.fs:26-55 normalises cross(world_b - world_t, world_l - world_r).  Where the two differences are parallel (a
    border texel whose missing neighbours fall back to its own depth, on a planar calibration volume) the products of the
    cross product cancel: exactly, to (0, 0, 0) and a NaN normal, when every product is rounded (the oracle, the HIP path:
    no FMA contraction); to a rounding residual -- normalised into an arbitrary unit vector -- when a*b - c*d is evaluated
    as fma(a, b, -(c*d)) (llvmpipe's JIT on a CPU with FMA).  Short of exact cancellation the direction is still dominated by
    the last bits of the looked-up positions.  The direction is undefined either way; such texels are recognised here in
    float64 by the conditioning of the cross product (below)."""
    import pyoracle
    H, W = depth_b.shape[:2]

    def outside(d):
        return d <= 0.0 or d >= 1.0

    def check(idx):
        y, x = idx
        d = float(depth_b[y, x, 0])
        if outside(d):
            return False
        world = {}
        for key, (dx, dy) in {"t": (0, 1), "b": (0, -1), "l": (-1, 0), "r": (1, 0)}.items():
            xx, yy = min(max(x + dx, 0), W - 1), min(max(y + dy, 0), H - 1)
            dn = float(depth_b[yy, xx, 0])
            dn = d if outside(dn) else dn
            world[key] = np.asarray(pyoracle.tex3d(scene.xyz[i], (x + dx + 0.5) / W, (y + dy + 0.5) / H, dn), np.float64)[:3]
        a, b = world["b"] - world["t"], world["l"] - world["r"]
        # the looked-up world positions carry ~1e-6 of absolute error (coordinates of magnitude 1, a trilinear lookup each);
        # the direction of cross(a, b) then carries about 1e-6 (|a| + |b|) (1 + |world|) / |cross|: beyond 1e-3 -- the loose
        # bound of the normal comparison -- the two runs need not agree at all
        wmax = max(float(np.abs(v).max()) for v in world.values())
        bound = 1e-6 * (float(np.linalg.norm(a)) + float(np.linalg.norm(b))) * (1.0 + wmax) / max(float(np.linalg.norm(np.cross(a, b))), 1e-300)
        return bound > 1e-3
    return check


def class_flips(t, r, ok, limit):
    """voxels whose class (-limit / inside the band / +limit) differs between the two volumes -- not counting a voxel
    where both values lie within 1e-6 of the same boundary (tsdf_integration.vs:41-46 compares sdist with +-limit: a
    last-bit difference of sdist turns exactly -limit into a weighted mean a hair above it; the values still agree
    to the TSDF tolerance, which is checked separately)"""
    limit = np.float32(limit)

    def cls(v):
        return np.where(v <= -limit, -1, np.where(v >= limit, 1, 0))
    d = (cls(t) != cls(r)) & ok
    at_boundary = (np.abs(np.abs(t) - limit) <= 1e-6) & (np.abs(np.abs(r) - limit) <= 1e-6) & (np.sign(t) == np.sign(r))
    return int((d & ~at_boundary).sum())


def grazing_angle_voxel(images, inv, res, limit, voxel, a, b, tol=0.0):
    """How far the weighted mean of tsdf_integration.vs:52 may move when every quality weight moves within the tolerance the
    quality images are compared with (TOL["quality"] absolute + RTOL["quality"] relative): sum |dw_i| |sd_i - mean| / sum w_i
    over the in-band sensors.  For ordinary voxels that is far below the TSDF tolerance.  It is not where all in-band sensors
    look at the surface at a grazing angle: quality = ... * angle^2 (pre_quality.fs:104-114) is 1e-5 ... 1e-17 there, its
    RELATIVE error is large or unbounded (angle = dot(view, normal) cancels to ~1e-6 and the normals of two runs differ by
    1e-5), and the mean may land anywhere between the sensors' signed distances.  True when |a - b| is within `tol` + twice
    that bound and both values lie inside the interval of the signed distances."""
    import pyoracle
    z, y, x = voxel
    lim = float(limit)
    lo, hi, wsum, terms = np.inf, -np.inf, 0.0, []
    for i in range(len(inv)):
        pc = pyoracle.tex3d(inv[i], (x + 0.5) / res[0], (y + 0.5) / res[1], (z + 0.5) / res[2])
        if pc[0] < 0:
            continue
        H, W = images["quality"][i].shape[:2]
        px, py = int(np.floor(pc[0] * W)), int(np.floor(pc[1] * H))
        sd = float(pc[2]) - float(np.asarray(images["depth_b"][i])[min(max(py, 0), H - 1), min(max(px, 0), W - 1), 0])
        if -lim < sd < lim:
            x0, y0 = int(np.floor(pc[0] * W - 0.5)), int(np.floor(pc[1] * H - 0.5))
            q = np.asarray(images["quality"][i])[max(y0, 0):y0 + 2, max(x0, 0):x0 + 2]
            w = float(np.nanmax(q)) if q.size else 0.0
            wsum += w
            terms.append((w, sd))
            lo, hi = min(lo, sd), max(hi, sd)
    if not terms or not (lo - 1e-6 <= min(a, b) and max(a, b) <= hi + 1e-6):
        return False
    if wsum < 1e-8:
        return True
    mean = 0.5 * (a + b)
    bound = sum((TOL["quality"] + RTOL["quality"] * w) * abs(sd - mean) for w, sd in terms) / wsum
    return abs(a - b) <= tol + 2.0 * bound


def compare(got, fx, name, what, scene, limit=0.01, counter_slack=0.0, inv=None):
    """`got`: images per sensor + counters + tsdf of the oracle or the HIP path; `fx`: the Mesa run"""
    n = shader_cases.MODE_CASES[name]["n"] if name in shader_cases.MODE_CASES else shader_cases.CASES[name][0]
    limit = np.float32(limit)
    u8 = name in shader_cases.COMPRESSED_DEPTH
    excused = {i: set() for i in range(n)}                 # degenerate-normal texels of the live random scenes, per sensor
    for k in shader_cases.IMAGES:
        for i in range(n):
            w = "%s vs Mesa: %s sensor %d" % (what, k, i)
            if k in EXACT and not (u8 and k in ("morph", "depth_rg")):
                assert same_bits(got[k][i], fx[k][i]), "%s: %d texels differ" % (w, count_diff(got[k][i], fx[k][i]))
            elif k in EXACT:
                within(got[k][i], fx[k][i], TOL_U8_DEPTH, w)
            else:
                g_, f_ = np.asarray(got[k][i], np.float32), np.asarray(fx[k][i], np.float32)
                if inv is not None and k in ("normal", "quality"):
                    # live random scenes: a texel whose normal is NaN on one side only has to be a degenerate one (see
                    # degenerate_normal); normal and quality of such texels are taken out of the comparison
                    gn = np.isnan(np.asarray(got["normal"][i], np.float32)).any(axis=-1)
                    fn = np.isnan(np.asarray(fx["normal"][i], np.float32)).any(axis=-1)
                    gnorm, fnorm = np.asarray(got["normal"][i], np.float32), np.asarray(fx["normal"][i], np.float32)
                    with np.errstate(invalid="ignore"):
                        apart = np.nan_to_num(np.abs(gnorm - fnorm).max(axis=-1)) > 1e-3      # (or two arbitrary unit vectors)
                    odd = np.argwhere((gn != fn) | apart)
                    assert len(odd) <= 8, "%s: %d texels with a NaN normal on one side only, or normals apart" % (w, len(odd))
                    is_degenerate = degenerate_normal(scene, i, np.asarray(got["depth_b"][i]))
                    g_, f_ = g_.copy(), f_.copy()
                    for y, x in odd:
                        assert is_degenerate((int(y), int(x))), "%s: NaN normal on one side only at %s, and the neighbourhood is not degenerate" % (w, (int(y), int(x)))
                        g_[y, x] = f_[y, x] = 0.0
                        excused[i].add((int(y), int(x)))
                within(g_, f_, TOL[k], w, allow_nan_in_want=(k == "quality"), rtol=RTOL.get(k, 0.0), loose=LOOSE.get(k),
                       excuse=negative_angle(scene, i, got["depth_b"][i], got["normal"][i]) if k == "quality" else None)
    dc = np.abs(got["counters"].astype(np.int64) - fx["counters"].astype(np.int64)).sum()
    # (live random scenes: a world position within an ulp of a brick face is counted in the neighbouring brick on one side
    # -- inc_bricks.glsl floors position / brick size --: 2 or 4 in the L1 difference per such pixel; a few pixels at most)
    assert dc <= max(counter_slack * fx["counters"].sum(), 8 if inv is not None else 0), "%s vs Mesa: brick counters differ by %d in total" % (what, dc)
    if "occupied" in getattr(fx, "files", fx) and "occupied" in got:      # updateOccupiedBricks' id list (bricks-on cases)
        assert np.array_equal(np.asarray(got["occupied"], np.uint32), fx["occupied"]), "%s vs Mesa: occupied bricks differ" % what
    t, r = np.asarray(got["tsdf"], np.float32), fx["tsdf"]
    # (5e-5 of the band's half-width: the bound scales with the limit, all committed cases use 0.01)
    tol = TOL_TSDF * max(1.0, float(limit) / 0.01)
    if inv is not None:
        tol *= 2.0           # live random scenes (thousands of them in a soak): the largest ordinary difference seen is 6.1e-7 at limit 0.01
    if inv is not None:
        # ... and a voxel that is NaN on THIS side only (Mesa's own NaNs are handled by `within`) has to weigh in one of those
        # texels' qualities (2 x 2 LINEAR footprint), or be a grazing-angle voxel whose weights underflow to exactly 0 here
        # (0 / 0) and to 1e-16 there
        import pyoracle
        t, r = t.copy(), np.array(r, np.float32, copy=True)
        res_ = t.shape[::-1]
        for z, y, x in np.argwhere(np.isnan(t) & ~np.isnan(r)):
            touched = False
            for i in range(n):
                pc = pyoracle.tex3d(inv[i], (x + 0.5) / res_[0], (y + 0.5) / res_[1], (z + 0.5) / res_[2])
                H_, W_ = np.asarray(got["quality"][i]).shape[:2]
                x0, y0 = int(np.floor(pc[0] * W_ - 0.5)), int(np.floor(pc[1] * H_ - 0.5))
                foot = {(min(max(y0 + dy, 0), H_ - 1), min(max(x0 + dx, 0), W_ - 1)) for dy in (0, 1) for dx in (0, 1)}
                touched = touched or bool(foot & excused[i])
            touched = touched or grazing_angle_voxel(got, inv, res_, limit, (int(z), int(y), int(x)), float(r[z, y, x]), float(r[z, y, x]))
            assert touched, "%s vs Mesa: voxel %s is NaN here only, touches no degenerate-normal texel and is no grazing-angle voxel" % (
                what, (int(x), int(y), int(z)))
            t[z, y, x] = r[z, y, x] = 0.0
    if inv is not None:      # live random scenes: a voxel beyond the bound has to be a grazing-angle one, checked per voxel
        far = np.argwhere(np.nan_to_num(np.abs(t.astype(np.float64) - r)) > tol)
        assert len(far) <= max(8, 5e-4 * t.size), "%s vs Mesa: %d voxels beyond %.3g" % (what, len(far), tol)      # (a cap; each is checked)
        t = t.copy()
        for z, y, x in far:
            assert grazing_angle_voxel(got, inv, t.shape[::-1], limit, (int(z), int(y), int(x)), float(t[z, y, x]), float(r[z, y, x]), tol), \
                "%s vs Mesa: voxel %s differs by %.3g and is no grazing-angle voxel" % (what, (int(x), int(y), int(z)), abs(float(t[z, y, x]) - float(r[z, y, x])))
            t[z, y, x] = r[z, y, x]
    skipped = within(t, r, tol, "%s vs Mesa: TSDF" % what, allow_nan_in_want=True)
    ok = ~(np.isnan(r) | np.isnan(t))

    flips = class_flips(t, r, ok, limit)
    assert flips == 0, "%s vs Mesa: %d voxels change class" % (what, flips)
    return skipped


def oracle_frame(orc, pkg, name):
    scene, cfg, geo, inv, inv_res = shader_cases.build(pkg.synth, pkg.capi, name)
    flags, G = shader_cases.CASES[name][5], shader_cases.CASES[name][3]
    ref = orc.run_pipeline(scene, pkg.synth.BBOX_MIN, pkg.synth.BBOX_MAX, (G, G, G), inv, limit=cfg.tsdf_limit,
                           brick_size=geo.brick_size, bv=geo.brick_voxels, res_bricks=tuple(geo.res_bricks),
                           filter_textures=bool(flags & 1), processed=bool(flags & 2), refine=bool(flags & 4), use_bricks=False,
                           compress=name in shader_cases.COMPRESSED_DEPTH)
    return scene, inv, ref


def test_the_vertex_stage_defect_of_llvmpipe_is_what_the_harness_pads_for():
    """Mesa against itself (a probe shader of the harness, no reference text, no oracle): after a fetch through a
    uniform-indexed sampler array the next fetch returns 0 in every vertex that is not the first of 8 -- the reason
    gl_ref.integrate draws one voxel centre per group of 8.  If a future Mesa fixes it the padding is merely unnecessary."""
    gl_ref = gl_lib()
    wrong = gl_ref.vs_sampler_array_bug(64)
    assert all(i % 8 != 0 for i in wrong), "a first-of-8 vertex is wrong as well: the padding does not cover this Mesa"


@pytest.mark.parametrize("name", sorted(shader_cases.CASES))
def test_mesa_runs_the_reference_glsl_and_reproduces_the_fixtures(pkg, name):
    gl_ref = gl_lib()
    import make_gl_golden
    scene, cfg, geo, inv, out = make_gl_golden.run_case(name)
    fx = fixture(name)
    assert bytes(fx["inputs_sha256"]).decode() == shader_cases.digest(scene, inv), "the synthetic scene drifted: regenerate the fixtures"
    for k in shader_cases.IMAGES:
        assert same_bits(np.stack(out[k]), fx[k]), "%s: Mesa no longer reproduces the committed fixture (%d differ)" % (k, count_diff(np.stack(out[k]), fx[k]))
    assert np.array_equal(out["counters"], fx["counters"])
    assert same_bits(out["tsdf"], fx["tsdf"])


@pytest.mark.parametrize("name", sorted(shader_cases.CASES))
def test_oracle_matches_the_reference_glsl_run(orc, pkg, name):
    scene, inv, ref = oracle_frame(orc, pkg, name)
    fx = fixture(name)
    assert bytes(fx["inputs_sha256"]).decode() == shader_cases.digest(scene, inv), "the synthetic scene drifted: regenerate the fixtures"
    compare(ref, fx, name, "oracle", scene)
    assert np.any(np.abs(fx["tsdf"]) < 0.01) and fx["counters"].sum() > 0          # the fixtures are not trivial


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(shader_cases.CASES))
def test_hip_path_matches_the_reference_glsl_run(pkg, name):
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
    got = {k: [ctx.readback_image(which, i) for i in range(n)] for k, which in IMG.items()}
    got["counters"] = ctx.readback_brick_counters()
    got["tsdf"] = ctx.readback_tsdf()
    ctx.close()
    compare(got, fx, name, "HIP path", scene)


# ---- consumers of the volume on Mesa: bricks.{vs,gs,fs} through the rasteriser with MIN blending (drawDepthLimits),
# ---- tsdf_raymarch.{vs,fs} over the unit cube with the depth test, framebuffer_transfer / tsdf_inpaint / tsdf_colorfill ----
TOL_PEEL = 1e-6            # rasterised gl_FragCoord.z against the analytic face depth
MAX_PEEL_EDGE = 2e-3       # fraction of peel values on a face edge where the rasteriser's fill rule decides
TOL_VIEW_DEPTH = 2e-5
TOL_VIEW_COLOR = 1e-2      # 8-bit bilinear weights of the colour frames, pow() of the Phong term on llvmpipe
TOL_FILL_COLOR = 2e-3


def view_fixture(name):
    return np.load(os.path.join(ROOT, "tests", "golden", "gl_views_%s.npz" % name))


def occupied_mask(orc, counters, min_voxels):
    ids, _ = orc.update_occupied(counters, min_voxels)
    mask = np.zeros(counters.shape, np.uint8)
    mask[ids] = 1
    return mask


def max_abs(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max()) if a.size else 0.0


@pytest.mark.parametrize("name", sorted(shader_cases.VIEW_CASES))
def test_mesa_reproduces_the_view_fixtures(pkg, name):
    gl_ref = gl_lib()
    import make_gl_golden
    scene, cfg, geo, inv, out = make_gl_golden.run_case(name, keep=True)
    views = gl_ref.run_views(name, scene, cfg, geo, inv, out)
    gl_ref.release(out)
    vf = view_fixture(name)
    assert sorted(views) == sorted(vf.files)
    for k in views:
        assert same_bits(views[k], vf[k]), "%s: Mesa no longer reproduces the committed fixture (%d differ)" % (k, count_diff(views[k], vf[k]))


@pytest.mark.parametrize("name", sorted(shader_cases.VIEW_CASES))
def test_oracle_views_match_the_reference_glsl_run(orc, pkg, name):
    """The oracle works on the frame Mesa produced (gl_passes fixture: volume, depth, quality, counters) and, for the
    space-skipping views, marches from Mesa's peel image, so every stage is compared on identical inputs:
      peels      oracle's per-ray grid walk == the rasterised, MIN-blended cube faces (incl. the geometry shader's cull)
      ray-march  same pixels hit, same number of samples per pixel, depth and colour within tolerance
      filling    the oracle fills Mesa's ray-marched frame; colours compared where the window's depth test (LESS
                 against the cleared 1.0, recon_integration.cpp:314 + kinect_client.cpp:614,994) lets the fragment
                 through, i.e. depth < 1 -- elsewhere the window keeps its clear colour"""
    scene, cfg, geo, inv, inv_res = shader_cases.build(pkg.synth, pkg.capi, name)
    fx, vf = fixture(name), view_fixture(name)
    n = scene.N
    mask = occupied_mask(orc, fx["counters"], cfg.min_voxels_per_brick)
    for key, eye, mode, skip, fill in shader_cases.VIEW_CASES[name]:
        view = shader_cases.make_view(pkg.capi, pkg.synth, eye, mode, skip)
        peels = None
        if skip:
            peels = vf[key + "_peels"]
            op = orc.depth_peels(bytes(view), pkg.synth.BBOX_MIN, geo.brick_size, tuple(geo.res_bricks), fx["counters"], mask)
            d = np.abs(op.astype(np.float64) - peels)
            assert (d > TOL_PEEL).mean() <= MAX_PEEL_EDGE, "%s: %d peel values differ from the rasteriser's" % (key, int((d > TOL_PEEL).sum()))
            assert np.array_equal(op[..., 0] < 1, peels[..., 0] < 1), "%s: peel coverage differs" % key
            assert 0.02 < (peels[..., 0] < 1).mean() < 0.98
        oc, od, on = orc.raymarch(bytes(view), fx["tsdf"], inv, scene.uv, [scene.color[i] for i in range(n)], list(fx["depth_b"]),
                                  list(fx["quality"]), limit=cfg.tsdf_limit, peels=peels)
        gc, gd, gn = vf[key + "_color"], vf[key + "_depth"], vf[key + "_samples"]
        assert np.array_equal(od < 1, gd < 1), "%s: %d pixels hit in one frame only" % (key, int(((od < 1) != (gd < 1)).sum()))
        assert 0.02 < (gd < 1).mean() < 0.98
        assert same_bits(on, gn), "%s: sample counts differ at %d pixels" % (key, count_diff(on, gn))
        assert max_abs(od, gd) <= TOL_VIEW_DEPTH, "%s: depth differs by %.3g" % (key, max_abs(od, gd))
        assert max_abs(oc, gc) <= TOL_VIEW_COLOR, "%s: colour differs by %.3g" % (key, max_abs(oc, gc))
        if fill:
            fc, fd = orc.fill_colors(gc, gd)
            gfc, gfd = vf[key + "_filled_color"], vf[key + "_filled_depth"]
            assert max_abs(fd, gfd) <= 1e-6
            shown = fd < 1
            assert np.all(gfc[~shown] == 0.0), "%s: a fragment with depth 1 passed the window's depth test" % key
            assert max_abs(fc[shown], gfc[shown]) <= TOL_FILL_COLOR, "%s: filled colour differs by %.3g" % (key, max_abs(fc[shown], gfc[shown]))
            assert (shown & (gc[..., 3] <= 0)).sum() > 0            # pixels were actually filled


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(shader_cases.VIEW_CASES))
def test_hip_views_match_the_reference_glsl_run(pkg, name):
    """The HIP path renders from its OWN frame (bit-identical to the oracle's, i.e. ulps away from Mesa's) and its own
    peels, so a ray may stop one sample earlier or later than on Mesa at a grazing pixel: the pixelwise comparisons
    allow 2 % of the pixels to differ, everything else is held to the tolerances of the oracle test."""
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
        if skip:   # k_depth_peels against the faces Mesa rasterised (the brick counters of the two runs are identical)
            hp, gp = ctx.draw_depth_limits(view), vf[key + "_peels"]
            assert (np.abs(hp.astype(np.float64) - gp) > TOL_PEEL).mean() <= MAX_PEEL_EDGE, "%s: peels differ from the rasteriser's" % key
            assert np.array_equal(hp[..., 0] < 1, gp[..., 0] < 1), "%s: peel coverage differs" % key
        color, depth, ns = ctx.raymarch(view)
        gc, gd, gn = vf[key + "_color"], vf[key + "_depth"], vf[key + "_samples"]
        npx = gd.size
        assert ((depth < 1) != (gd < 1)).sum() <= 0.02 * npx, "%s: coverage" % key
        assert (ns != gn).sum() <= 0.02 * npx, "%s: %d sample counts differ" % (key, int((ns != gn).sum()))
        both = (depth < 1) & (gd < 1)
        assert (np.abs(depth - gd)[both] > TOL_VIEW_DEPTH).sum() <= 0.02 * npx, "%s: depth" % key
        assert (np.abs(color - gc)[both].max(axis=-1) > TOL_VIEW_COLOR).sum() <= 0.02 * npx, "%s: colour" % key
        if fill:
            fc, fd = ctx.fill_colors(view.width, view.height)
            gfc, gfd = vf[key + "_filled_color"], vf[key + "_filled_depth"]
            shown = (fd < 1) & (gfd < 1)
            assert ((fd < 1) != (gfd < 1)).sum() <= 0.02 * npx
            assert (np.abs(fc - gfc)[shown].max(axis=-1) > TOL_VIEW_COLOR).sum() <= 0.03 * npx, "%s: filled colour" % key
    ctx.close()


# ---- beyond the committed cases (build container only: Mesa runs live) -----------------------------------------------
def live_compare(orc, pkg, gl_ref, scene, cfg, geo, res, inv, what, bricks=False, counter_slack=0.0, **kw):
    """Mesa and the oracle on the same scene; the tolerances of `compare`.  With bricks=True the integration draws the
    index lists of the occupied bricks (ReconIntegration::integrate with m_use_bricks, the reference's default;
    recon_integration.cpp:255-261) instead of every voxel."""
    n = scene.N
    args = (scene, pkg.synth.BBOX_MIN, pkg.synth.BBOX_MAX, res, inv)
    common = dict(limit=cfg.tsdf_limit, brick_size=geo.brick_size, res_bricks=tuple(geo.res_bricks), **kw)
    got = gl_ref.run_frame(*args, keep=bricks, **common)
    ref = orc.run_pipeline(*args, bv=geo.brick_voxels, use_bricks=bricks, **common)
    if bricks:
        # the voxels of the occupied bricks: divideBox + containedVoxels (host code) on MESA's counters
        ids, _ = orc.update_occupied(got["counters"], cfg.min_voxels_per_brick)
        occ = np.zeros(got["counters"].shape, np.uint8)
        occ[ids] = 1
        vmask, _, _ = orc.brick_voxel_mask(pkg.synth.BBOX_MIN, pkg.synth.BBOX_MAX, geo.brick_size, res, occ)
        st = got["_gl"]
        H, W = scene.depth.shape[1:3]
        got["tsdf"] = gl_ref.integrate(st["cal"], st["tex"], n, res, cfg.tsdf_limit, (W, H), indices=[np.nonzero(vmask.ravel())[0]])
        gl_ref.release(got)
    fx = {k: np.stack(got[k]) for k in shader_cases.IMAGES}
    fx["counters"], fx["tsdf"] = got["counters"], got["tsdf"]

    name = "__live__"
    shader_cases.CASES[name] = (n,)
    try:
        compare(ref, fx, name, what, scene, limit=cfg.tsdf_limit, counter_slack=counter_slack, inv=inv)
    finally:
        del shader_cases.CASES[name]
    return fx, ref


@pytest.mark.parametrize("seed", range(6 + int(os.environ.get("RGBDR_EXTRA_MESA_SEEDS", "0"))))
def test_random_small_scenes_oracle_matches_mesa(orc, pkg, seed):
    """random sensor counts (1-5), image / LUT / grid sizes (mostly not powers of two), host toggles and limits; every
    second seed integrates through the occupied bricks' index lists like the reference's default mode"""
    gl_ref = gl_lib()
    rng = np.random.default_rng(2000 + seed)
    n = int(rng.integers(1, 5)) if seed else 5       # seed 0: all five slots of the shaders' sampler3D[5] arrays
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
    live_compare(orc, pkg, gl_ref, scene, cfg, geo, res, inv, "oracle (seed %d)" % seed, bricks=bool(seed % 2),
                 filter_textures=bool(flags & 1), processed=bool(flags & 2), refine=bool(flags & 4))


@pytest.mark.parametrize("seed", range(3 + int(os.environ.get("RGBDR_EXTRA_MESA_SEEDS", "0")) // 4))
def test_random_views_oracle_matches_mesa(orc, pkg, seed):
    """the consumer side live: a random small scene through Mesa, then two random views of it -- viewport size, eye (outside /
    inside the box), shade mode, with and without the brick depth peels, hole filling -- through bricks.{vs,gs,fs},
    tsdf_raymarch.{vs,fs} and the inpaint pyramid on Mesa, against the oracle working on Mesa's own frame (volume, images,
    counters, peel image): same pixels hit, same sample counts, depth / colour / filled colour within the view tolerances.
    A ray whose first positive sample is within rounding of zero may stop one sample apart on the two sides: at most a
    pixel or two per view, counted."""
    gl_ref = gl_lib()
    import pyoracle
    rng = np.random.default_rng(9000 + seed)
    n = int(rng.integers(1, 5))
    wh = (int(rng.integers(24, 49)), int(rng.integers(20, 41)))
    lut_res = tuple(int(v) for v in rng.integers(8, 15, 3))
    G = int(rng.choice([16, 20, 24, 32]))
    limit = float(rng.choice([0.01, 0.02, 0.035]))
    scene = pkg.synth.Scene(n, wh[0], wh[1], lut_res=lut_res, seed=int(rng.integers(1, 10000)))
    cfg = pkg.capi.make_config(n, wh, voxel_size=2.0 / G, brick_size=8 * 2.0 / G, tsdf_limit=limit, min_voxels=int(rng.choice([1, 10])))
    geo = pkg.capi.compute_geometry(cfg)
    res = tuple(geo.res_volume)
    inv = scene.inverse(res)
    out = gl_ref.run_frame(scene, pkg.synth.BBOX_MIN, pkg.synth.BBOX_MAX, res, inv, limit=limit, brick_size=geo.brick_size,
                           res_bricks=tuple(geo.res_bricks), keep=True)
    try:
        st = out["_gl"]
        mask = occupied_mask(orc, out["counters"], cfg.min_voxels_per_brick)
        eyes = [(2.2, 1.6, 1.9), (0.85, 1.7, 0.8), (-2.0, 1.2, 2.1), (0.05, 1.95, 0.02)]
        for _ in range(2):
            # (aspect ratios of a window: at 58 x 8 -- 147 degrees across -- the rasteriser's interpolated ray positions and
            # the per-pixel ones of the oracle put ceil(|t_far - t_near|) one sample apart for one ray in twenty)
            vh = int(rng.integers(16, 48))
            vw = int(rng.integers(vh, min(2 * vh, 80)))
            view = pkg.capi.make_view(eyes[int(rng.integers(0, 4))], (0.0, 0.9, 0.0), (0.0, 1.0, 0.0), 50.0, vw, vh, pkg.synth.BBOX_MIN,
                                      pkg.synth.BBOX_MAX, shade_mode=int(rng.integers(0, 4)))
            view.skip_space = int(rng.integers(0, 2)) if mask.any() else 0
            what = "seed %d view %dx%d mode %d skip %d" % (seed, vw, vh, view.shade_mode, view.skip_space)
            v = pyoracle.View.from_buffer_copy(bytes(view))
            peels, peels_tex = None, None
            if view.skip_space:
                peels, peels_tex = gl_ref.depth_limits(v, geo.brick_size, tuple(geo.res_bricks), out["counters"], cfg.min_voxels_per_brick)
                op = orc.depth_peels(bytes(view), pkg.synth.BBOX_MIN, geo.brick_size, tuple(geo.res_bricks), out["counters"], mask)
                d = np.abs(op.astype(np.float64) - peels)
                # (pixels on a face edge, where the rasteriser's fill rule decides: a share of the image at 96 x 72, a number
                # proportional to the perimeter in viewports of a few hundred pixels)
                # viewports of a few hundred pixels put a brick face on three or four of them: the rasteriser's interpolated
                # gl_FragCoord.z carries up to 1e-5 there (1e-6 at 96 x 72, the committed cases' TOL_PEEL)
                assert (d > 10 * TOL_PEEL).sum() <= max(MAX_PEEL_EDGE * d.size, 2 * (vw + vh)), "%s: %d peel values differ from the rasteriser's" % (
                    what, int((d > 10 * TOL_PEEL).sum()))
                # coverage: a pixel whose centre lies on a cube's silhouette within rounding may fall on either side -- only
                # pixels of the coverage boundary (a 4-neighbour covered differently), and few of them
                co, cm = op[..., 0] < 1, peels[..., 0] < 1
                pad = np.pad(cm, 1, mode="edge")
                boundary = (pad[:-2, 1:-1] != cm) | (pad[2:, 1:-1] != cm) | (pad[1:-1, :-2] != cm) | (pad[1:-1, 2:] != cm)
                assert not ((co != cm) & ~boundary).any(), "%s: peel coverage differs inside a face" % what
                assert (co != cm).sum() <= max(2, 0.03 * cm.size), "%s: peel coverage differs at %d pixels" % (what, int((co != cm).sum()))
            gc, gd, gn, target = gl_ref.raymarch(v, st["cal"], st["tex"], st["volume"], n, limit, peels_tex)
            oc, od, on = orc.raymarch(bytes(view), out["tsdf"], inv, scene.uv, [scene.color[i] for i in range(n)], list(out["depth_b"]),
                                      list(out["quality"]), limit=limit, peels=peels)
            apart = (od < 1) != (gd < 1)
            with np.errstate(invalid="ignore"):
                apart |= ~((on == gn) | (np.isnan(on) & np.isnan(gn)))
            assert apart.sum() <= max(2, 2e-3 * apart.size), "%s: %d pixels stop at different samples" % (what, int(apart.sum()))
            ok = ~apart
            assert max_abs(od[ok], gd[ok]) <= TOL_VIEW_DEPTH, "%s: depth differs by %.3g" % (what, max_abs(od[ok], gd[ok]))
            fin = ok[..., None] & np.isfinite(oc) & np.isfinite(gc)
            assert max_abs(oc[fin], gc[fin]) <= TOL_VIEW_COLOR, "%s: colour differs by %.3g" % (what, max_abs(oc[fin], gc[fin]))
            gfc, gfd, _ = gl_ref.fill_colors(target, v.width, v.height)
            fc, fd = orc.fill_colors(gc, gd)                    # the oracle fills Mesa's ray-marched frame
            assert max_abs(fd, gfd) <= 1e-6, what
            shown = (fd < 1) & np.isfinite(fc).all(axis=-1) & np.isfinite(gfc).all(axis=-1)
            assert np.all(gfc[~(fd < 1)] == 0.0), "%s: a fragment with depth 1 passed the window's depth test" % what
            if view.shade_mode != 2:
                # (mode 2 draws normals: negative red channels, which tsdf_inpaint.fs:52-75 takes for its "no sample" marker
                # -- a window of such samples averages 0 / 0, and what the pyramid makes of the NaN from there on differs
                # between llvmpipe's comparisons and IEEE's: two pixels of one view in fifty seeds; filled normals are not a
                # thing the reference shows)
                # (5 * TOL_FILL_COLOR = the ray-march's colour tolerance: the blend of tsdf_colorfill.fs:44-60 divides by w1 + w2, llvmpipe's
                # reciprocal; 5.5e-3 is the largest difference in 900 random views, 5.3e-4 in the committed ones)
                assert max_abs(fc[shown], gfc[shown]) <= 5 * TOL_FILL_COLOR, "%s: filled colour differs by %.3g" % (what, max_abs(fc[shown], gfc[shown]))
            if peels_tex is not None:
                gl_ref.delete_textures([peels_tex])
    finally:
        gl_ref.release(out)


def test_full_sensor_resolution_oracle_matches_mesa(orc, pkg):
    """BASELINE's sensor size: two 512 x 424 sensors (13 x 13 bilateral over 217 088 pixels each, the scene's 0.5 m sphere)
    into a 64^3 volume, every pass on Mesa against the oracle"""
    gl_ref = gl_lib()
    G = 64
    scene = pkg.synth.Scene(2, 512, 424, lut_res=(32, 27, 32), seed=4321)
    cfg = pkg.capi.make_config(2, (512, 424), voxel_size=2.0 / G, brick_size=8 * 2.0 / G)
    geo = pkg.capi.compute_geometry(cfg)
    inv = scene.inverse((G, G, G))
    # 434 176 marked positions, some on the scene's floor plane = the box's y = 0 face: a last-bit difference of the
    # LINEAR position fetch moves such a point across a brick boundary or out of the grid (where the shader converts a
    # negative float to uvec3 -- undefined; the oracle skips the point, llvmpipe counts it somewhere): 76 of 194 114 counts
    fx, ref = live_compare(orc, pkg, gl_ref, scene, cfg, geo, (G, G, G), inv, "oracle (512 x 424)", counter_slack=1e-3,
                           filter_textures=True, processed=True, refine=True)
    assert (np.abs(fx["tsdf"]) < cfg.tsdf_limit).sum() > 1000 and fx["counters"].sum() > 10000


@pytest.mark.parametrize("mode", [1, 5])
def test_dxt_blocks_decode_within_one_step_of_the_gl_drivers_decode(orc, pkg, mode):
    """The reference hands its DXT colour frames to the GL driver (GL_COMPRESSED_RGBA_S3TC_DXT1 / DXT5 layers); the
    library and the oracle decode them with squish's integer arithmetic -- the reference's own CPU decoder of the same
    frames (NetKinectArray.cpp:633).  EXT_texture_compression_s3tc leaves the rounding of the two interpolated palette
    entries to the implementation: llvmpipe evaluates (2 a + b) / 3 as a + ((b - a) * 85 >> 8) and the DXT1 midpoint as
    (a + b + 1) >> 1, squish truncates the exact quotient.  So, on random blocks (both end-point orders) and on an
    encoded picture: the end-point colours (indices 0 and 1) and transparent black are identical, the interpolated
    entries within one step of 255."""
    gl_ref = gl_lib()
    rng = np.random.default_rng(77 + mode)
    W, H = 64, 40
    nb = (W // 4) * (H // 4)
    blocks = rng.integers(0, 256, (nb, 8 if mode == 1 else 16), dtype=np.uint8)
    blocks[: nb // 8, 2:4] = blocks[: nb // 8, 0:2]                    # some blocks with equal end points
    o = 0 if mode == 1 else 8
    for blk in (blocks, pkg.synth.encode_dxt(rng.integers(0, 256, (H, W, 3), dtype=np.uint8), mode)):
        blk = np.ascontiguousarray(blk, np.uint8).reshape(nb, -1)
        want = gl_ref.decode_dxt(blk, W, H, mode)[..., :3].astype(np.int32)
        got = orc.decode_dxt(blk, W, H, mode).astype(np.int32)
        assert np.abs(got - want).max() <= 1, "more than one step from Mesa's S3TC decode"
        yy, xx = np.mgrid[0:H, 0:W]
        b = blk[(yy // 4) * (W // 4) + xx // 4].astype(np.int32)                          # [H, W, bytes]
        rows = np.take_along_axis(b[..., o + 4:o + 8], (yy & 3)[..., None], axis=-1)[..., 0]   # the texel's index byte
        idx = (rows >> (2 * (xx & 3))) & 3
        c0 = b[..., o] | (b[..., o + 1] << 8)
        c1 = b[..., o + 2] | (b[..., o + 3] << 8)
        exact = (idx < 2) | ((mode == 1) & (c0 <= c1) & (idx == 3))                      # end points, transparent black
        assert np.array_equal(got[exact], want[exact]), "an end-point colour differs from Mesa's decode"
        assert exact.any() and (~exact).any()


def test_u8_depth_through_the_morph_pass_oracle_matches_mesa(orc, pkg):
    """The combination SURVEY A.5 calls incoherent -- u8 depth frames AND processed depth: pre_morph.fs tests
    0.5 < d < 4.5 on the raw [0, 1] values (:19-26), pre_depth.fs un-compresses the morph output (:63-72) -- is still
    defined behaviour; Mesa and the oracle agree on it"""
    gl_ref = gl_lib()
    G = 32
    scene = pkg.synth.Scene(2, 64, 53, lut_res=(16, 14, 16), seed=21)
    scene.depth_u8 = pkg.synth.compress_depth_u8(scene.depth)
    scene.depth = (scene.depth_u8.astype(np.float32) / np.float32(255.0)).astype(np.float32)
    cfg = pkg.capi.make_config(2, (64, 53), voxel_size=2.0 / G, brick_size=8 * 2.0 / G, compress_depth=1)
    geo = pkg.capi.compute_geometry(cfg)
    inv = scene.inverse((G, G, G))
    shader_cases.COMPRESSED_DEPTH.add("__live__")
    try:
        fx, ref = live_compare(orc, pkg, gl_ref, scene, cfg, geo, (G, G, G), inv, "oracle (u8 + morph)", compress=True,
                               filter_textures=True, processed=True, refine=True)
    finally:
        shader_cases.COMPRESSED_DEPTH.discard("__live__")
    assert fx["counters"].sum() > 0 and (fx["morph"] != 0).sum() > 1000


# ---- BASELINE's sensor size, committed as a sample: four 512 x 424 sensors into 128^3 on Mesa ------------------------
def sample_fixture():
    return np.load(os.path.join(ROOT, "tests", "golden", "gl_sample_four_sensors_512x424_into_128.npz"))


def sample_scene(pkg):
    G = 128
    scene = pkg.synth.Scene(4, 512, 424, lut_res=(32, 27, 32), seed=1234)
    cfg = pkg.capi.make_config(4, (512, 424), voxel_size=2.0 / G, brick_size=8 * 2.0 / G)
    geo = pkg.capi.compute_geometry(cfg)
    inv = scene.inverse((G, G, G))
    return scene, cfg, geo, inv, G


def compare_sample(got, fx, scene, limit, what, inv=None):
    """`got`: full images per sensor, counters and volume of the oracle / the HIP path; `fx`: the sampled Mesa run
    (tests/golden/make_gl_golden.py: 19 814 texels of every image, 59 413 voxels -- half of them in the surface band)"""
    n, H, W = 4, 424, 512
    tex = fx["texels"].astype(np.int64)
    si, rem = np.divmod(tex, H * W)
    ty, tx = np.divmod(rem, W)
    full = {k: np.stack([np.asarray(a, np.float32) for a in got[k]]) for k in shader_cases.IMAGES}
    for k in shader_cases.IMAGES:
        g = full[k].reshape(n * H * W, -1)[tex]
        w = "%s vs Mesa (4 x 512 x 424): %s" % (what, k)
        if k in EXACT:
            assert same_bits(g, fx[k]), "%s: %d sampled values differ" % (w, count_diff(g, fx[k]))
        elif k == "quality":
            want = fx[k]
            llvm_nan = np.isnan(want) & ~np.isnan(g)
            assert llvm_nan.mean() <= MAX_NAN_FRACTION
            for j in np.flatnonzero(llvm_nan.reshape(-1)):
                i = int(si[j])
                assert negative_angle(scene, i, full["depth_b"][i], full["normal"][i])((int(ty[j]), int(tx[j]))), \
                    "%s: NaN on llvmpipe at sensor %d (%d, %d) without a negative angle or a NaN neighbour" % (w, i, ty[j], tx[j])
            ok = ~llvm_nan
            within(g[ok], want[ok], TOL[k], w, rtol=RTOL[k])
        else:
            within(g, fx[k], TOL[k], w, loose=LOOSE.get(k))
    dc = np.abs(np.asarray(got["counters"]).astype(np.int64) - fx["counters"].astype(np.int64)).sum()
    assert dc <= 1e-3 * fx["counters"].sum(), "%s: brick counters differ by %d of %d" % (what, dc, fx["counters"].sum())
    t = np.asarray(got["tsdf"], np.float32).reshape(-1)[fx["voxels"].astype(np.int64)]
    r = fx["tsdf"]
    far = np.flatnonzero(np.abs(t.astype(np.float64) - r) > TOL_TSDF)
    if far.size and inv is not None:
        # A voxel all of whose in-band sensors look at the surface at a grazing angle: quality = ... * angle^2 (pre_quality.fs:
        # 104-114) is ~1e-13 there and its RELATIVE error is unbounded (angle = dot(view, normal) cancels to ~1e-6 and the
        # normals of the two runs differ by 1e-5), so the weighted mean of tsdf_integration.vs:52 may land anywhere between
        # the sensors' signed distances.  Each such voxel is checked: total weight < 1e-8, both values inside that interval.
        assert far.size <= 2e-4 * t.size, "%s: %d sampled voxels beyond %.3g" % (what, far.size, TOL_TSDF)
        G = inv[0].shape[0]
        for j in far:
            z, rem = divmod(int(fx["voxels"][j]), G * G)
            y, x = divmod(rem, G)
            lo, hi, wsum = np.inf, -np.inf, 0.0
            for i in range(len(inv)):
                pc = inv[i][z, y, x]                                   # 1:1 with the grid: the texel itself
                if pc[0] < 0:
                    continue
                px, py = int(np.floor(pc[0] * W)), int(np.floor(pc[1] * H))
                sd = float(pc[2]) - float(full["depth_b"][i][min(max(py, 0), H - 1), min(max(px, 0), W - 1), 0])
                if -limit < sd < limit:
                    x0, y0 = int(np.floor(pc[0] * W - 0.5)), int(np.floor(pc[1] * H - 0.5))
                    q = full["quality"][i][max(y0, 0):y0 + 2, max(x0, 0):x0 + 2]
                    wsum += float(np.nanmax(q))
                    lo, hi = min(lo, sd), max(hi, sd)
            assert wsum < 1e-8, "%s: voxel %s differs by %.3g with total weight %.3g" % (what, (x, y, z), abs(float(t[j]) - float(r[j])), wsum)
            assert lo - 1e-6 <= min(t[j], r[j]) and max(t[j], r[j]) <= hi + 1e-6, "%s: voxel %s outside its sensors' distances" % (what, (x, y, z))
        keep = np.ones(t.size, bool)
        keep[far] = False
        t, r = t[keep], r[keep]
    within(t, r, TOL_TSDF, "%s vs Mesa (4 x 512 x 424): TSDF" % what, allow_nan_in_want=True)
    ok = ~(np.isnan(r) | np.isnan(t))
    lim = np.float32(limit)

    flips = class_flips(t, r, ok, lim)
    assert flips == 0, "%s: %d sampled voxels change class" % (what, flips)
    assert (np.abs(r[ok]) < lim).sum() > 20000


def test_oracle_matches_the_mesa_sample_at_baseline_sensor_size(orc, pkg):
    scene, cfg, geo, inv, G = sample_scene(pkg)
    fx = sample_fixture()
    assert bytes(fx["inputs_sha256"]).decode() == shader_cases.digest(scene, inv), "the synthetic scene drifted: regenerate the fixture"
    ref = orc.run_pipeline(scene, pkg.synth.BBOX_MIN, pkg.synth.BBOX_MAX, (G, G, G), inv, limit=cfg.tsdf_limit, brick_size=geo.brick_size,
                           bv=geo.brick_voxels, res_bricks=tuple(geo.res_bricks), use_bricks=False)
    compare_sample(ref, fx, scene, cfg.tsdf_limit, "oracle")


@pytest.mark.gpu
def test_hip_path_matches_the_mesa_sample_at_baseline_sensor_size(pkg):
    capi = pkg.capi
    scene, cfg, geo, inv, G = sample_scene(pkg)
    fx = sample_fixture()
    assert bytes(fx["inputs_sha256"]).decode() == shader_cases.digest(scene, inv)
    ctx = capi.Context(cfg, 0)
    for i in range(4):
        ctx.set_calibration(i, scene.xyz[i], scene.lut_res, scene.uv[i], scene.lut_res, (0.5, 4.5))
        ctx.set_inverse_calibration(i, inv[i], (G, G, G))
    ctx.set_use_bricks(False)
    ctx.step(scene.depth, scene.color)
    got = {k: [ctx.readback_image(which, i) for i in range(4)] for k, which in IMG.items()}
    got["counters"] = ctx.readback_brick_counters()
    got["tsdf"] = ctx.readback_tsdf()
    ctx.close()
    compare_sample(got, fx, scene, cfg.tsdf_limit, "HIP path")


# ---- the reference's DEFAULT mode and its limits on Mesa (round 4) -----------------------------------------------------------
# bricks on: updateOccupiedBricks on Mesa's own counters, then one glDrawElements per occupied brick with the brick's
# containedVoxels list (gl_ref.host_grid / brick_indices restate divideBox + containedVoxels; recon_integration.cpp:255-261) --
# on a power-of-two grid, on the reference's own box with 5-voxel bricks that share rows, and on a grid whose last x brick
# lists an index past the axis end (aliasing through the linear index); DXT1 colour layers decoded by the GL
# (NetKinectArray.cpp:149-156) next to squish's decode in the oracle / the library; five sensors (sampler3D[5]).
MODES = sorted(shader_cases.MODE_CASES)
TOL_LAB_DXT = 6e-3         # + one step of 255 in the interpolated palette entries (Mesa's S3TC decode vs squish), 2.5e-3 seen


def mode_frame_kwargs(name, cfg, geo):
    f = shader_cases.MODE_CASES[name]["flags"]
    return dict(limit=cfg.tsdf_limit, brick_size=geo.brick_size, res_bricks=tuple(geo.res_bricks), filter_textures=bool(f & 1),
                processed=bool(f & 2), refine=bool(f & 4), use_bricks=bool(f & 8))


def compare_mode(got, fx, name, what, scene, limit):
    if shader_cases.MODE_CASES[name].get("dxt"):
        TOL["lab"], keep = TOL_LAB_DXT, TOL["lab"]
        try:
            return compare(got, fx, name, what, scene, limit=limit)
        finally:
            TOL["lab"] = keep
    return compare(got, fx, name, what, scene, limit=limit)


@pytest.mark.parametrize("name", MODES)
def test_mesa_reproduces_the_default_mode_fixtures(pkg, name):
    gl_lib()
    import make_gl_golden
    scene, cfg, geo, inv, out = make_gl_golden.run_mode_case(name)
    fx = fixture(name)
    assert bytes(fx["inputs_sha256"]).decode() == shader_cases.digest_mode(scene, inv), "the synthetic scene drifted: regenerate the fixtures"
    for k in shader_cases.IMAGES:
        assert same_bits(np.stack(out[k]), fx[k]), "%s: Mesa no longer reproduces the committed fixture" % k
    assert np.array_equal(out["counters"], fx["counters"]) and np.array_equal(out["occupied"], fx["occupied"])
    assert same_bits(out["tsdf"], fx["tsdf"])


def test_the_harness_brick_lists_are_the_reference_construction(pkg, orc):
    """gl_ref.host_grid / brick_indices (the harness's own restatement of setVoxelSize, setBrickSize, divideBox and
    containedVoxels, which decides WHICH voxel centres Mesa draws in the bricks-on cases) against the oracle's literal
    nested loops (orc_brick_voxel_mask) and the library's tables -- three independent restatements of the same host code"""
    import gl_ref
    for name in MODES:
        c = shader_cases.MODE_CASES[name]
        cfg = pkg.capi.make_config(c["n"], c["wh"], bbox_min=c["bbox"][0], bbox_max=c["bbox"][1], voxel_size=c["voxel"], brick_size=c["brick"])
        geo = pkg.capi.compute_geometry(cfg)
        grid = gl_ref.host_grid(c["bbox"][0], c["bbox"][1], c["voxel"], c["brick"])
        assert grid["res"] == tuple(geo.res_volume) and grid["res_bricks"] == tuple(geo.res_bricks) and grid["brick_size"] == geo.brick_size
        assert grid["res_bricks"] == orc.divide_box(c["bbox"][0], c["bbox"][1], geo.brick_size)
        X, Y, Z = grid["res"]
        rng = np.random.default_rng(3)
        occ = (rng.random(geo.num_bricks) < 0.1).astype(np.uint8)
        occ[grid["res_bricks"][0] - 1] = 1                          # a brick at the x end: the one that overflows, if any does
        mine = np.zeros(X * Y * Z, np.uint8)
        for b in np.flatnonzero(occ):
            ids = gl_ref.brick_indices(grid, int(b))
            mine[ids[ids < X * Y * Z]] = 1
        lit, _, _ = orc.brick_voxel_mask(c["bbox"][0], c["bbox"][1], geo.brick_size, grid["res"], occ)
        assert np.array_equal(mine.reshape(Z, Y, X), lit), "%s: %d voxels differ" % (name, int((mine.reshape(Z, Y, X) != lit).sum()))
    g = gl_ref.host_grid(*shader_cases.MODE_CASES["bricks_last_brick_overflows_the_axis"]["bbox"], 0.03, 0.15)
    assert g["axes"][0][-1][1] > g["res"][0], "the overflow case no longer overflows"


def test_the_harness_camera_positions_are_the_reference_construction(pkg, orc):
    """gl_ref.host_camera_pos (CalibVolumes::getCameraPositions = Frustum::getCameraPos of the cv_xyz corner samples, the
    uniform pre_quality.fs gets on Mesa) is the harness's own binary32 restatement; it agrees with the oracle's C
    restatement bit for bit, with the library's rgbdr_camera_position, and with the analytic sensor position to 1e-6"""
    import gl_ref
    for n, wh, lut, seed in ((4, (128, 106), (32, 27, 32), 1234), (5, (64, 53), (16, 14, 16), 4242), (3, (48, 40), (12, 10, 12), 5)):
        scene = pkg.synth.Scene(n, wh[0], wh[1], lut_res=lut, seed=seed, make_frames=False)
        for i in range(n):
            a = gl_ref.host_camera_pos(scene.xyz[i])
            assert same_bits(a, orc.camera_pos(scene.xyz[i]))
            assert same_bits(a, np.asarray(pkg.capi.camera_position(scene.xyz[i], scene.lut_res), np.float32))
            assert np.abs(a - scene.sensors[i].pos).max() < 1e-6


def oracle_mode_frame(orc, pkg, name):
    c = shader_cases.MODE_CASES[name]
    scene, cfg, geo, inv, inv_res = shader_cases.build_mode(pkg.synth, pkg.capi, name, decode_dxt=orc.decode_dxt)
    ref = orc.run_pipeline(scene, c["bbox"][0], c["bbox"][1], tuple(geo.res_volume), inv, bv=geo.brick_voxels,
                           min_voxels=cfg.min_voxels_per_brick, **mode_frame_kwargs(name, cfg, geo))
    return scene, cfg, inv, ref


@pytest.mark.parametrize("name", MODES)
def test_oracle_matches_the_reference_glsl_run_in_its_default_mode(orc, pkg, name):
    scene, cfg, inv, ref = oracle_mode_frame(orc, pkg, name)
    fx = fixture(name)
    assert bytes(fx["inputs_sha256"]).decode() == shader_cases.digest_mode(scene, inv), "the synthetic scene drifted: regenerate the fixtures"
    compare_mode(ref, fx, name, "oracle", scene, cfg.tsdf_limit)
    t = fx["tsdf"]
    assert np.any(np.abs(t) < cfg.tsdf_limit) and 0 < fx["occupied"].size < fx["counters"].size
    # bricks on: most of the volume is the cleared -limit of voxels no occupied brick lists
    assert (t == -np.float32(cfg.tsdf_limit)).mean() > 0.5


@pytest.mark.gpu
@pytest.mark.parametrize("name", MODES)
def test_hip_path_matches_the_reference_glsl_run_in_its_default_mode(pkg, name):
    """the library in the reference's default mode (RGBDR_FLAG_USE_BRICKS set; DXT1 blocks handed over as they are)"""
    capi = pkg.capi
    c = shader_cases.MODE_CASES[name]
    scene, cfg, geo, inv, inv_res = shader_cases.build_mode(pkg.synth, capi, name)
    fx = fixture(name)
    assert bytes(fx["inputs_sha256"]).decode() == shader_cases.digest_mode(scene, inv)
    n = c["n"]
    ctx = capi.Context(cfg, 0)
    for i in range(n):
        ctx.set_calibration(i, scene.xyz[i], scene.lut_res, scene.uv[i], scene.lut_res, (0.5, 4.5))
        ctx.set_inverse_calibration(i, inv[i], inv_res)
    assert ctx.cfg.flags & 8
    ctx.step(scene.depth, scene.color_blocks if c.get("dxt") else scene.color)
    got = {k: [ctx.readback_image(which, i) for i in range(n)] for k, which in IMG.items()}
    got["counters"] = ctx.readback_brick_counters()
    got["occupied"] = ctx.get_occupied()[0]
    got["tsdf"] = ctx.readback_tsdf()
    ctx.close()
    compare_mode(got, fx, name, "HIP path", scene, cfg.tsdf_limit)


# ---- larger samples: BASELINE's grids (256^3 swept whole, z bands of the 512^3 headline grid) and the default mode (DXT1
# ---- 1280 x 1080 colour + bricks on) at BASELINE's sensor size; tests/golden/make_gl_golden.py BIG_SAMPLES ----------------------
def big_fixture(name):
    return np.load(os.path.join(ROOT, "tests", "golden", "gl_sample_%s.npz" % name))


def big_inputs(pkg, name, decode_dxt=None):
    import make_gl_golden as mg
    scene, cfg, geo, inv = mg.big_scene(name, decode_dxt=decode_dxt)
    fx = big_fixture(name)
    assert bytes(fx["inputs_sha256"]).decode() == mg.big_digest(scene, inv, name), "the synthetic scene drifted: regenerate the fixture"
    return mg.BIG_SAMPLES[name], scene, cfg, geo, inv, fx


def compare_big(got, fx, scene, cfg, c, what, inv=None):
    if c.get("dxt"):
        TOL["lab"], keep = TOL_LAB_DXT, TOL["lab"]
    try:
        compare_sample(got, fx, scene, cfg.tsdf_limit, what, inv=inv)
    finally:
        if c.get("dxt"):
            TOL["lab"] = keep
    if "occupied" in fx.files:
        assert np.array_equal(np.asarray(got["occupied"], np.uint32), fx["occupied"]), "%s: occupied bricks differ from Mesa's" % what


BIG = ["four_sensors_512x424_into_256", "four_sensors_512x424_into_512_bands", "default_mode_dxt1_bricks_512x424_into_128"]


@pytest.mark.parametrize("name", BIG)
def test_oracle_matches_the_large_mesa_samples(orc, pkg, name):
    c, scene, cfg, geo, inv, fx = big_inputs(pkg, name, decode_dxt=orc.decode_dxt)
    G = c["G"]
    kw = dict(limit=cfg.tsdf_limit, brick_size=geo.brick_size, bv=geo.brick_voxels, res_bricks=tuple(geo.res_bricks),
              min_voxels=cfg.min_voxels_per_brick)
    if "bands" in c:
        ref = orc.run_pipeline(scene, pkg.synth.BBOX_MIN, pkg.synth.BBOX_MAX, (G, G, G), None, use_bricks=False, **kw)
        vol = np.full((G, G, G), np.nan, np.float32)
        for z0, z1 in c["bands"]:
            # the oracle reads RGBA records: the band's rows with a zero fourth component, at their place in a grid-sized view
            luts = []
            for a in inv:
                full = np.zeros((G, G, G, 4), np.float32)                 # calloc: untouched pages cost nothing
                full[z0:z1, ..., :a.shape[-1]] = a[z0:z1]
                luts.append(full)
            orc.integrate(luts, ref["sil"], ref["depth_b"], ref["quality"], (G, G, G), cfg.tsdf_limit, z_range=(z0, z1), out=vol)
        ref["tsdf"] = vol
    else:
        ref = orc.run_pipeline(scene, pkg.synth.BBOX_MIN, pkg.synth.BBOX_MAX, (G, G, G), inv, use_bricks=bool(c.get("bricks")), **kw)
    compare_big(ref, fx, scene, cfg, c, "oracle", inv=inv)


@pytest.mark.gpu
@pytest.mark.parametrize("name", BIG)
def test_hip_path_matches_the_large_mesa_samples(pkg, name):
    capi = pkg.capi
    c, scene, cfg, geo, inv, fx = big_inputs(pkg, name)
    G = c["G"]
    ctx = capi.Context(cfg, 0)
    for i in range(4):
        ctx.set_calibration(i, scene.xyz[i], scene.lut_res, scene.uv[i], scene.lut_res, (0.5, 4.5))
        a = inv[i]
        if a.shape[-1] == 3:                                              # the library takes the file's RGBA32F records
            full = np.zeros((G, G, G, 4), np.float32)
            for z0, z1 in c["bands"]:
                full[z0:z1, ..., :3] = a[z0:z1]
            a = full
        ctx.set_inverse_calibration(i, a, (G, G, G))
        del a
    assert bool(ctx.cfg.flags & 8) == bool(c.get("bricks"))
    ctx.step(scene.depth, scene.color_blocks if c.get("dxt") else scene.color)
    got = {k: [ctx.readback_image(which, i) for i in range(4)] for k, which in IMG.items()}
    got["counters"] = ctx.readback_brick_counters()
    got["occupied"] = ctx.get_occupied()[0]
    got["tsdf"] = ctx.readback_tsdf()
    ctx.close()
    compare_big(got, fx, scene, cfg, c, "HIP path", inv=inv)
