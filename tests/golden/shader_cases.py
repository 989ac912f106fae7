"""The scenes behind tests/golden/shader_passes_*.npz (shared by make_shader_golden.py and tests/test_shader_ref.py).
Deterministic from the seeds: the inputs are regenerated wherever the fixtures are used, and a digest of them is stored
with the outputs so that a drift of the synthetic scene shows up as such."""
import hashlib

import numpy as np

# name -> (sensors, (W, H), forward LUT res, grid G, inverse LUT res or None for 1:1, flags, seed)
# flags: 1 filter_textures, 2 processed depth, 4 refine boundary (rgbdr_config.flags / NetKinectArray's toggles)
CASES = {
    "two_sensors_default": (2, (64, 53), (16, 14, 16), 32, None, 7, 1234),
    "three_sensors_coarse_inverse_lut": (3, (64, 53), (16, 14, 16), 32, (23, 25, 22), 7, 77),
    "no_filter_raw_depth_no_refine": (2, (48, 40), (12, 10, 12), 24, None, 0, 5),
    "filter_only_fine_inverse_lut": (2, (48, 40), (12, 10, 12), 24, (30, 30, 30), 1, 9),
    "four_sensors_128x106_into_64": (4, (128, 106), (32, 27, 32), 64, None, 7, 1234),
    # u8 depth frames (compress_depth: pre_depth.fs un-compresses; the reference's u8 + morph combination is incoherent,
    # SURVEY A.5, so raw depth) and a grid that is not a power of two (voxel centres are not texel centres of a 1:1 LUT)
    "u8_depth_raw_non_pow2_grid": (2, (64, 53), (16, 14, 16), 40, None, 5, 21),
}
COMPRESSED_DEPTH = {"u8_depth_raw_non_pow2_grid"}
IMAGES = ("morph", "depth_rg", "lab", "depth_b", "sil", "normal", "quality")


def build(pkg_synth, capi, name):
    n, wh, lut_res, G, inv_res, flags, seed = CASES[name]
    scene = pkg_synth.Scene(n, wh[0], wh[1], lut_res=lut_res, seed=seed)
    compress = name in COMPRESSED_DEPTH
    if compress:                   # the frames as the server sends them (u8) and as the raw-depth texture holds them ([0, 1])
        scene.depth_u8 = pkg_synth.compress_depth_u8(scene.depth)
        scene.depth = (scene.depth_u8.astype(np.float32) / np.float32(255.0)).astype(np.float32)
    cfg = capi.make_config(n, wh, voxel_size=2.0 / G, brick_size=8 * 2.0 / G, flags=flags | 8, compress_depth=1 if compress else 0)
    geo = capi.compute_geometry(cfg)
    inv_res = inv_res or (G, G, G)
    inv = scene.inverse(inv_res)
    return scene, cfg, geo, inv, inv_res


def digest(scene, inv):
    h = hashlib.sha256()
    for a in (scene.depth, scene.color, *scene.xyz, *scene.uv, *inv):
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


# ---- consumers of the volume: views ray-marched (and hole-filled) from the frames of the cases above -----------------
# case -> [(key, eye, shade mode, skip_space, hole filling)]; viewport 48 x 36, target (0, 0.9, 0), fov 50
VIEW_CASES = {
    "two_sensors_default": [("outside_m0", (2.2, 1.6, 1.9), 0, 0, False), ("outside_m1", (2.2, 1.6, 1.9), 1, 0, False),
                            ("outside_m2", (2.2, 1.6, 1.9), 2, 0, False), ("outside_m3", (2.2, 1.6, 1.9), 3, 0, False),
                            ("inside_m0", (0.85, 1.7, 0.8), 0, 0, True), ("outside_skip_fill", (2.2, 1.6, 1.9), 0, 1, True)],
    "four_sensors_128x106_into_64": [("outside_skip_fill", (2.2, 1.6, 1.9), 0, 1, True), ("inside_m1", (0.85, 1.7, 0.8), 1, 0, False)],
}
VIEWPORT = (48, 36)


def make_view(capi, synth, eye, mode, skip):
    v = capi.make_view(eye, (0.0, 0.9, 0.0), (0.0, 1.0, 0.0), 50.0, VIEWPORT[0], VIEWPORT[1], synth.BBOX_MIN, synth.BBOX_MAX, shade_mode=mode)
    v.skip_space = skip
    return v
