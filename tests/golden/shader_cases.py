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


# ---- the reference's DEFAULT mode and its limits (round 4): tests/golden/gl_passes_<name>.npz, Mesa run only ---------------
# bricks on = ReconIntegration::integrate draws the occupied bricks' index lists (m_use_bricks{true},
# recon_integration.cpp:55,255-261; kinect_client.cpp:82,278); compress_rgb 1 = DXT1 colour layers decoded by the GL
# (NetKinectArray.cpp:149-156, the yml default KinectCalibrationFile.cpp:88-95); 5 sensors = every slot of the shaders'
# sampler3D[5] arrays (tsdf_integration.vs:13).  Keys: n, wh, lut (forward LUT res), bbox, voxel, brick, inv (inverse LUT res
# or None for 1:1 with the grid), flags (1 filter, 2 processed, 4 refine, 8 bricks), seed, dxt (compress_rgb), color_wh
BOX = ((-1.0, 0.0, -1.0), (1.0, 2.0, 1.0))
MODE_CASES = {
    # power-of-two grid: bricks coincide with the library's 8^3 storage tiles
    "bricks_two_sensors_pow2_grid": dict(n=2, wh=(64, 53), lut=(16, 14, 16), bbox=BOX, voxel=2.0 / 32, brick=8 * 2.0 / 32, inv=None,
                                         flags=15, seed=1234),
    # the reference's own box and brick / voxel ratio, scaled: (-1,0,-1)-(1,2.2,1), voxel 0.02, bricks of 0.1 m = 5 voxels
    # (100 x 111 x 100 voxels -- the y resolution one more than 2.2 / 0.02, as 221 is at the default voxel size --, 20 x 22 x 20
    # bricks most of which also list the first row of the next brick), inverse LUT at the calib_inverter ratio 0.7
    "bricks_reference_box_5_voxel_bricks": dict(n=3, wh=(128, 106), lut=(32, 27, 32), bbox=((-1.0, 0.0, -1.0), (1.0, 2.2, 1.0)),
                                                voxel=0.02, brick=0.1, inv=(143, 158, 143), flags=15, seed=77),
    # the last brick of x and of y lists an index past the axis end (61 of 61 voxels): the linear index z*X*Y + y*X + x
    # aliases a voxel of the next row / slice; the box cuts the scene's sphere, so those bricks are occupied
    "bricks_last_brick_overflows_the_axis": dict(n=2, wh=(128, 106), lut=(32, 27, 32), bbox=((-1.0, 0.0, -1.0), (0.82, 1.81, 1.0)),
                                                 voxel=0.03, brick=0.15, inv=(43, 43, 47), flags=15, seed=5),
    # the default mode in full: DXT1 colour (at a colour resolution != depth resolution) + bricks on
    "dxt1_colour_bricks_on": dict(n=2, wh=(64, 53), lut=(16, 14, 16), bbox=BOX, voxel=2.0 / 32, brick=8 * 2.0 / 32, inv=None,
                                  flags=15, seed=1234, dxt=1, color_wh=(96, 80)),
    "five_sensors_bricks_on": dict(n=5, wh=(64, 53), lut=(16, 14, 16), bbox=BOX, voxel=2.0 / 32, brick=8 * 2.0 / 32, inv=None,
                                   flags=15, seed=4242),
}


def build_mode(pkg_synth, capi, name, decode_dxt=None):
    """-> scene, cfg, geo, inv, inv_res.  A DXT case carries scene.color_blocks (what the server sends: the GL run and the
    HIP path take these) and, with decode_dxt = the oracle's squish restatement, scene.color = the decoded RGB8 frames
    (what the oracle's pre_depth takes)."""
    c = MODE_CASES[name]
    scene = pkg_synth.Scene(c["n"], c["wh"][0], c["wh"][1], lut_res=c["lut"], seed=c["seed"], color_wh=c.get("color_wh"))
    dxt = c.get("dxt", 0)
    cfg = capi.make_config(c["n"], c["wh"], color_wh=c.get("color_wh"), bbox_min=c["bbox"][0], bbox_max=c["bbox"][1],
                           voxel_size=c["voxel"], brick_size=c["brick"], flags=c["flags"], compress_rgb=dxt)
    geo = capi.compute_geometry(cfg)
    if dxt:
        scene.color_blocks = np.stack([pkg_synth.encode_dxt(scene.color[i], dxt) for i in range(c["n"])])
        if decode_dxt is not None:
            wc, hc = c.get("color_wh") or c["wh"]
            scene.color = np.stack([decode_dxt(scene.color_blocks[i], wc, hc, dxt) for i in range(c["n"])])
    inv_res = c["inv"] or tuple(geo.res_volume)
    inv = scene.inverse(inv_res, c["bbox"][0], c["bbox"][1])
    return scene, cfg, geo, inv, inv_res


def digest_mode(scene, inv):
    h = hashlib.sha256()
    for a in (scene.depth, getattr(scene, "color_blocks", scene.color), *scene.xyz, *scene.uv, *inv):
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()
