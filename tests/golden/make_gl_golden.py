#!/usr/bin/env python3
"""Writes tests/golden/gl_passes_<case>.npz (and gl_views_<case>.npz): what the reference's GLSL -- glsl/pre_*.fs,
tsdf_integration.vs, ... read where they lie under /root/reference and RUN by Mesa llvmpipe (oracle/gl_ref.py,
oracle/gl_context.c; build container only) -- produces for the scenes of shader_cases.py.  The fixtures are data
(images, counters, volumes, frames); no reference text is stored.

These are outputs of the reference's own shader code executed by a real OpenGL implementation (compiler, texture units,
rasteriser, image stores, atomics), with the reference's host-side GL state restated by gl_ref.py.  They are what pins
the oracle's pass arithmetic (tests/test_gl_ref.py: oracle vs these fixtures on CPU, HIP path vs these fixtures on GPU).

    make -C oracle glctx && python tests/golden/make_gl_golden.py [case ...]"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), HERE]
from __graft_entry__ import load_oracle, load_package  # noqa: E402

load_oracle()
load_package()
import gl_ref  # noqa: E402
import shader_cases  # noqa: E402
from rgbd_recon_amd import capi, synth  # noqa: E402


def frame_kwargs(name, cfg, geo):
    flags = shader_cases.CASES[name][5]
    return dict(limit=cfg.tsdf_limit, brick_size=geo.brick_size, res_bricks=tuple(geo.res_bricks), filter_textures=bool(flags & 1),
                processed=bool(flags & 2), refine=bool(flags & 4), compress=name in shader_cases.COMPRESSED_DEPTH)


def run_case(name, keep=False):
    scene, cfg, geo, inv, inv_res = shader_cases.build(synth, capi, name)
    G = shader_cases.CASES[name][3]
    out = gl_ref.run_frame(scene, synth.BBOX_MIN, synth.BBOX_MAX, (G, G, G), inv, keep=keep, **frame_kwargs(name, cfg, geo))
    return scene, cfg, geo, inv, out


def main(argv):
    assert gl_ref.available(), "oracle/_ref/libglctx.so or /root/reference/glsl is missing: make -C oracle glctx"
    names = argv or list(shader_cases.CASES)
    info = gl_ref.info()
    print("running the reference's GLSL on", info["renderer"], "/", info["version"])
    bug = gl_ref.vs_sampler_array_bug()
    print("llvmpipe vertex-shader sampler-array defect present:", bug)
    for name in names:
        scene, cfg, geo, inv, out = run_case(name, keep=True)
        n = scene.N
        G = shader_cases.CASES[name][3]
        # the padding that works around llvmpipe's vertex-shader defect must not matter: one centre per 8 == one per 16
        again = gl_ref.integrate(out["_gl"]["cal"], out["_gl"]["tex"], n, (G, G, G), cfg.tsdf_limit, shader_cases.CASES[name][1], pad=16)
        assert np.array_equal(again.view(np.uint32), out["tsdf"].view(np.uint32)), "padding 8 and 16 disagree"
        arrays = {k: np.stack(out[k]) for k in shader_cases.IMAGES}
        arrays["counters"] = out["counters"]
        arrays["tsdf"] = out["tsdf"]
        arrays["inputs_sha256"] = np.frombuffer(shader_cases.digest(scene, inv).encode(), dtype=np.uint8)
        arrays["gl_renderer"] = np.frombuffer((info["renderer"] + " / " + info["version"]).encode(), dtype=np.uint8)
        path = os.path.join(HERE, "gl_passes_%s.npz" % name)
        np.savez_compressed(path, **arrays)
        print("%-40s %7.1f KiB  surface voxels %d  counted %d  NaN qualities %d" % (
            name, os.path.getsize(path) / 1024, int(np.sum(np.abs(out["tsdf"]) < cfg.tsdf_limit)), int(out["counters"].sum()),
            int(np.isnan(arrays["quality"]).sum())))
        if hasattr(gl_ref, "run_views") and name in shader_cases.VIEW_CASES:
            views = gl_ref.run_views(name, scene, cfg, geo, inv, out)
            path = os.path.join(HERE, "gl_views_%s.npz" % name)
            np.savez_compressed(path, **views)
            print("%-40s %7.1f KiB  %s" % ("  views", os.path.getsize(path) / 1024,
                  ", ".join("%s: %d px hit" % (k[:-6], int((v < 1).sum())) for k, v in views.items() if k.endswith("_depth") and "filled" not in k)))
        gl_ref.release(out)




# ---- the reference's default mode and its limits: bricks-on index lists, DXT1 colour layers, five sensors ----------------
def mode_kwargs(name, cfg, geo):
    c = shader_cases.MODE_CASES[name]
    f = c["flags"]
    return dict(limit=cfg.tsdf_limit, brick_size=geo.brick_size, res_bricks=tuple(geo.res_bricks), filter_textures=bool(f & 1),
                processed=bool(f & 2), refine=bool(f & 4), use_bricks=bool(f & 8), min_voxels=cfg.min_voxels_per_brick,
                compress_rgb=c.get("dxt", 0))


def run_mode_case(name, keep=False):
    scene, cfg, geo, inv, inv_res = shader_cases.build_mode(synth, capi, name)
    c = shader_cases.MODE_CASES[name]
    # setVoxelSize / setBrickSize / divideBox restated by the harness itself (gl_ref.host_grid) must agree with the library's
    grid = gl_ref.host_grid(c["bbox"][0], c["bbox"][1], c["voxel"], c["brick"])
    assert grid["res"] == tuple(geo.res_volume) and grid["res_bricks"] == tuple(geo.res_bricks) and grid["brick_size"] == geo.brick_size
    out = gl_ref.run_frame(scene, c["bbox"][0], c["bbox"][1], tuple(geo.res_volume), inv, keep=keep, **mode_kwargs(name, cfg, geo))
    return scene, cfg, geo, inv, out


def main_modes(names):
    info = gl_ref.info()
    for name in names:
        scene, cfg, geo, inv, out = run_mode_case(name)
        arrays = {k: np.stack(out[k]) for k in shader_cases.IMAGES}
        arrays["counters"], arrays["tsdf"] = out["counters"], out["tsdf"]
        if "occupied" in out:
            arrays["occupied"] = out["occupied"]
        arrays["inputs_sha256"] = np.frombuffer(shader_cases.digest_mode(scene, inv).encode(), dtype=np.uint8)
        arrays["gl_renderer"] = np.frombuffer((info["renderer"] + " / " + info["version"]).encode(), dtype=np.uint8)
        path = os.path.join(HERE, "gl_passes_%s.npz" % name)
        np.savez_compressed(path, **arrays)
        t = out["tsdf"]
        print("%-40s %7.1f KiB  grid %s  surface voxels %d  +limit voxels %d  occupied bricks %d of %d  counted %d" % (
            name, os.path.getsize(path) / 1024, "x".join(str(v) for v in geo.res_volume), int(np.sum(np.abs(t) < cfg.tsdf_limit)),
            int(np.sum(t >= cfg.tsdf_limit)), len(out.get("occupied", [])), out["counters"].size, int(out["counters"].sum())))


# ---- BASELINE's sensor size: four 512 x 424 sensors into 128^3, frozen as a SAMPLE (the full frame would be 60 MB) ----
SAMPLE_NAME = "four_sensors_512x424_into_128"
SAMPLE_G = 128
SAMPLE_TEXELS, SAMPLE_VOXELS = 20000, 60000


def sample_scene():
    G = SAMPLE_G
    scene = synth.Scene(4, 512, 424, lut_res=(32, 27, 32), seed=1234)
    cfg = capi.make_config(4, (512, 424), voxel_size=2.0 / G, brick_size=8 * 2.0 / G)
    geo = capi.compute_geometry(cfg)
    inv = scene.inverse((G, G, G))
    return scene, cfg, geo, inv


def make_sample():
    """the Mesa run of the whole frame; stored: every brick counter, SAMPLE_TEXELS texels of every image and SAMPLE_VOXELS
    voxels (half of them drawn from the surface band) at seeded positions"""
    scene, cfg, geo, inv = sample_scene()
    G = SAMPLE_G
    out = gl_ref.run_frame(scene, synth.BBOX_MIN, synth.BBOX_MAX, (G, G, G), inv, limit=cfg.tsdf_limit, brick_size=geo.brick_size,
                           res_bricks=tuple(geo.res_bricks), filter_textures=True, processed=True, refine=True)
    rng = np.random.default_rng(4242)
    n, H, W = 4, 424, 512
    arrays = {"counters": out["counters"], "inputs_sha256": np.frombuffer(shader_cases.digest(scene, inv).encode(), dtype=np.uint8)}
    # texels: half uniformly, half where the sensor saw something (depth_b.r in (0, 1))
    db = np.stack(out["depth_b"])[..., 0]
    seen = np.flatnonzero((db > 0) & (db < 1))
    tex = np.unique(np.concatenate([rng.integers(0, n * H * W, SAMPLE_TEXELS // 2), rng.choice(seen, SAMPLE_TEXELS // 2, replace=False)]))
    arrays["texels"] = tex.astype(np.uint32)
    for k in shader_cases.IMAGES:
        a = np.stack(out[k])
        arrays[k] = a.reshape(n * H * W, -1)[tex]
    t = out["tsdf"].reshape(-1)
    band = np.flatnonzero(np.abs(t) < cfg.tsdf_limit)
    vox = np.unique(np.concatenate([rng.integers(0, t.size, SAMPLE_VOXELS // 2), rng.choice(band, min(SAMPLE_VOXELS // 2, band.size), replace=False)]))
    arrays["voxels"] = vox.astype(np.uint32)
    arrays["tsdf"] = t[vox]
    info = gl_ref.info()
    arrays["gl_renderer"] = np.frombuffer((info["renderer"] + " / " + info["version"]).encode(), dtype=np.uint8)
    path = os.path.join(HERE, "gl_sample_%s.npz" % SAMPLE_NAME)
    np.savez_compressed(path, **arrays)
    print("%-40s %7.1f KiB  %d texels, %d voxels (%d in the band of %d), counted %d" % (
        SAMPLE_NAME, os.path.getsize(path) / 1024, tex.size, vox.size, int((np.abs(t[vox]) < cfg.tsdf_limit).sum()), band.size, int(out["counters"].sum())))

# ---- larger samples (round 4): BASELINE's grids and the reference's default mode at BASELINE's sensor size ---------------
# name -> G (grid), full sweep / z bands / bricks, colour format.  Stored like the 128^3 sample: every brick counter, seeded
# texel and voxel samples (half of the voxels from the surface band).
BIG_SAMPLES = {
    # BASELINE configs[1]'s grid, swept whole by tsdf_integration.vs (16.7 M voxel centres)
    "four_sensors_512x424_into_256": dict(G=256),
    # BASELINE configs[2]'s grid = the headline.  llvmpipe refuses the 2 GiB RGBA32F inverse-LUT texture of a 512^3 grid, so
    # this one sample stores the LUT texels as RGB32F (the fourth component is never read: tsdf_integration.vs:31 takes
    # .xyz; make_big_sample checks on the 128^3 scene that both formats give the same volume bit for bit) and runs the
    # shader for the voxel centres of four z bands (rows outside the bands hold zeros in the LUT and are not drawn)
    "four_sensors_512x424_into_512_bands": dict(G=512, bands=((0, 8), (200, 208), (252, 260), (504, 512)), rgb_only=True),
    # the reference's default mode at BASELINE's sensor size: DXT1 colour 1280 x 1080 decoded by the GL, bricks on
    "default_mode_dxt1_bricks_512x424_into_128": dict(G=128, dxt=1, color_wh=(1280, 1080), bricks=True),
}


def big_scene(name, decode_dxt=None, lut_rows_only=True):
    """-> scene, cfg, geo, inv (a list of [Z,Y,X,4] -- or [Z,Y,X,3] with zeros outside the bands for the banded sample)"""
    c = BIG_SAMPLES[name]
    G = c["G"]
    scene = synth.Scene(4, 512, 424, lut_res=(32, 27, 32), seed=1234, color_wh=c.get("color_wh"))
    cfg = capi.make_config(4, (512, 424), color_wh=c.get("color_wh"), voxel_size=2.0 / G, brick_size=8 * 2.0 / G,
                           compress_rgb=c.get("dxt", 0), flags=15 if c.get("bricks") else 7)
    geo = capi.compute_geometry(cfg)
    if c.get("dxt"):
        scene.color_blocks = np.stack([synth.encode_dxt(scene.color[i], c["dxt"]) for i in range(4)])
        if decode_dxt is not None:
            scene.color = np.stack([decode_dxt(scene.color_blocks[i], c["color_wh"][0], c["color_wh"][1], c["dxt"]) for i in range(4)])
    if "bands" in c:
        inv = []
        for s in scene.sensors:
            a = np.zeros((G, G, G, 3 if c.get("rgb_only") else 4), np.float32)
            for z0, z1 in c["bands"]:
                a[z0:z1] = synth.inverse_lut(s, (G, G, G), z_range=(z0, z1))[..., :a.shape[-1]]
            inv.append(a)
    else:
        inv = scene.inverse((G, G, G))
    return scene, cfg, geo, inv


def band_voxels(name):
    c = BIG_SAMPLES[name]
    G = c["G"]
    return np.concatenate([np.arange(z0 * G * G, z1 * G * G, dtype=np.int64) for z0, z1 in c["bands"]])


def big_digest(scene, inv, name):
    import hashlib
    h = hashlib.sha256()
    for a in (scene.depth, getattr(scene, "color_blocks", scene.color), *scene.xyz, *scene.uv):
        h.update(np.ascontiguousarray(a).tobytes())
    c = BIG_SAMPLES[name]
    for a in inv:                                    # the LUT rows the sample can see (hashing 8 GB would take a minute)
        for z0, z1 in c.get("bands", ((0, 4), (c["G"] // 2, c["G"] // 2 + 4))):
            h.update(np.ascontiguousarray(a[z0:z1]).tobytes())
    return h.hexdigest()


def make_big_sample(name):
    c = BIG_SAMPLES[name]
    G = c["G"]
    if c.get("rgb_only"):          # RGB32F against RGBA32F inverse-LUT textures on a grid both fit: the same volume, bit for bit
        sc, cf, ge, iv = sample_scene()
        kw = dict(limit=cf.tsdf_limit, brick_size=ge.brick_size, res_bricks=tuple(ge.res_bricks))
        a = gl_ref.run_frame(sc, synth.BBOX_MIN, synth.BBOX_MAX, (SAMPLE_G,) * 3, iv, **kw)["tsdf"]
        b = gl_ref.run_frame(sc, synth.BBOX_MIN, synth.BBOX_MAX, (SAMPLE_G,) * 3, iv, inv_rgb_only=True, **kw)["tsdf"]
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "RGB32F and RGBA32F inverse LUTs disagree on Mesa"
        print("RGB32F and RGBA32F inverse-LUT textures give the same 128^3 volume bit for bit")
    scene, cfg, geo, inv = big_scene(name)
    vox_ids = band_voxels(name) if "bands" in c else None
    out = gl_ref.run_frame(scene, synth.BBOX_MIN, synth.BBOX_MAX, (G, G, G), inv, limit=cfg.tsdf_limit, brick_size=geo.brick_size,
                           res_bricks=tuple(geo.res_bricks), filter_textures=True, processed=True, refine=True,
                           compress_rgb=c.get("dxt", 0), use_bricks=bool(c.get("bricks")), min_voxels=cfg.min_voxels_per_brick,
                           inv_rgb_only=bool(c.get("rgb_only")), integrate_voxels=vox_ids)
    rng = np.random.default_rng(4242)
    n, H, W = 4, 424, 512
    arrays = {"counters": out["counters"], "inputs_sha256": np.frombuffer(big_digest(scene, inv, name).encode(), dtype=np.uint8)}
    if "occupied" in out:
        arrays["occupied"] = out["occupied"]
    db = np.stack(out["depth_b"])[..., 0]
    seen = np.flatnonzero((db > 0) & (db < 1))
    edge = np.flatnonzero(np.stack(out["depth_b"])[..., 1] > 0)          # pre_boundary's edge classes (q = 0.1 / 1.0): all of them
    tex = np.unique(np.concatenate([rng.integers(0, n * H * W, SAMPLE_TEXELS // 2), rng.choice(seen, SAMPLE_TEXELS // 2, replace=False),
                                    edge[:SAMPLE_TEXELS]]))
    arrays["texels"] = tex.astype(np.uint32)
    for k in shader_cases.IMAGES:
        arrays[k] = np.stack(out[k]).reshape(n * H * W, -1)[tex]
    t = out["tsdf"].reshape(-1)
    pool = vox_ids if vox_ids is not None else None
    tt = t if pool is None else t[pool]
    band = np.flatnonzero(np.abs(tt) < cfg.tsdf_limit)
    pick = np.unique(np.concatenate([rng.integers(0, tt.size, SAMPLE_VOXELS // 2), rng.choice(band, min(SAMPLE_VOXELS // 2, band.size), replace=False)]))
    vox = pick if pool is None else pool[pick]
    arrays["voxels"] = vox.astype(np.uint32)
    arrays["tsdf"] = t[vox]
    info = gl_ref.info()
    arrays["gl_renderer"] = np.frombuffer((info["renderer"] + " / " + info["version"]).encode(), dtype=np.uint8)
    path = os.path.join(HERE, "gl_sample_%s.npz" % name)
    np.savez_compressed(path, **arrays)
    print("%-46s %7.1f KiB  %d texels (%d edge classes), %d voxels (%d in the band of %d), counted %d, occupied bricks %s" % (
        name, os.path.getsize(path) / 1024, tex.size, edge.size, vox.size, int((np.abs(t[vox]) < cfg.tsdf_limit).sum()), band.size,
        int(out["counters"].sum()), len(out["occupied"]) if "occupied" in out else "-"))


if __name__ == "__main__":
    args = sys.argv[1:]
    if args == ["sample"]:
        make_sample()
    elif args and all(a in shader_cases.MODE_CASES for a in args):
        main_modes(args)
    elif args and all(a in BIG_SAMPLES for a in args):
        for a in args:
            make_big_sample(a)
    elif args == ["big"]:
        for a in BIG_SAMPLES:
            make_big_sample(a)
    elif args == ["modes"]:
        main_modes(list(shader_cases.MODE_CASES))
    else:
        main(args)
        if not args:
            main_modes(list(shader_cases.MODE_CASES))
            make_sample()
            for a in BIG_SAMPLES:
                make_big_sample(a)
