"""ctypes bindings of the CPU oracle (oracle/rgbdr_oracle.c) -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py import
this module.  The product path (rgbd-recon_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# RGBDR_ORACLE_LIB: a differently built copy of the oracle (tests/mutation_check.py loads its mutants that way)
LIB_PATH = os.environ.get("RGBDR_ORACLE_LIB") or os.path.join(HERE, "librgbdr_oracle.so")
REF_PATH = os.path.join(HERE, "_ref", "libref_shim.so")
_F = C.POINTER(C.c_float)


class DepthParams(C.Structure):
    _fields_ = [("W", C.c_int), ("H", C.c_int), ("Wc", C.c_int), ("Hc", C.c_int),
                ("xyz_res", C.c_int * 3), ("uv_res", C.c_int * 3),
                ("cv_min_ds", C.c_float), ("cv_max_ds", C.c_float),
                ("bbox_min", C.c_float * 3), ("bbox_max", C.c_float * 3),
                ("filter_textures", C.c_int), ("compress", C.c_int),
                ("near_", C.c_float), ("far_", C.c_float)]


class NormalParams(C.Structure):
    _fields_ = [("W", C.c_int), ("H", C.c_int), ("xyz_res", C.c_int * 3),
                ("bbox_min", C.c_float * 3), ("bbox_max", C.c_float * 3),
                ("brick_size", C.c_float), ("res_bricks", C.c_int * 3)]


class IntegrateParams(C.Structure):
    _fields_ = [("num_sensors", C.c_int), ("W", C.c_int), ("H", C.c_int), ("res", C.c_int * 3),
                ("limit", C.c_float)]


class View(C.Structure):
    """orc_view (same layout as rgbdr_view)"""
    _fields_ = [("modelview", C.c_float * 16), ("projection", C.c_float * 16), ("normal_matrix", C.c_float * 16),
                ("gl_normal_matrix_inv", C.c_float * 16), ("vol_to_world", C.c_float * 16),
                ("vol_to_world_inv", C.c_float * 16), ("modelview_inv", C.c_float * 16), ("img_to_eye", C.c_float * 16),
                ("camera_pos", C.c_float * 3), ("width", C.c_int), ("height", C.c_int), ("shade_mode", C.c_int),
                ("skip_space", C.c_int)]


class RaymarchParams(C.Structure):
    _fields_ = [("num_sensors", C.c_int), ("W", C.c_int), ("H", C.c_int), ("Wc", C.c_int), ("Hc", C.c_int),
                ("res", C.c_int * 3), ("limit", C.c_float)]


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", HERE, "portable"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        l = C.CDLL(LIB_PATH)
        l.orc_pow6.restype = C.c_float
        l.orc_pow6.argtypes = [C.c_float]
        l.orc_pow2.restype = C.c_float
        l.orc_pow2.argtypes = [C.c_float]
        l.orc_adjust_brick_size.restype = C.c_float
        l.orc_adjust_brick_size.argtypes = [C.c_float, C.c_float]
        l.orc_volume_sampler_resize.restype = C.c_double
        l.orc_tex3d_linear.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                       C.c_float, C.c_void_p]
        l.orc_tex2d_linear.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p]
        l.orc_axis_nearest.argtypes = [C.c_float, C.c_int]
        l.orc_volume_res.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
        l.orc_divide_box.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
        l.orc_brick_voxel_mask.restype = C.c_size_t
        l.orc_brick_voxel_mask.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                           C.c_void_p, C.c_void_p]
        l.orc_update_occupied.restype = C.c_uint32
        l.orc_update_occupied.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        _lib = l
    return _lib


def ref_lib():
    """oracle/_ref shim built from the reference's own sources (None if absent)."""
    if not os.path.exists(REF_PATH):
        return None
    l = C.CDLL(REF_PATH)
    return l


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def set_threads(n):
    return lib().orc_set_threads(int(n))


def set_linear_weight_bits(bits):
    """driver-tolerance study only (oracle/driver_weight_bound.py): LINEAR weights rounded to `bits` fractional bits, 0 = exact"""
    lib().orc_set_linear_weight_bits(int(bits))


def linear_weight_bits():
    return int(lib().orc_get_linear_weight_bits())


def tex3d(vol, u, v, w):
    vol = f32(vol)
    rz, ry, rx, ch = vol.shape
    out = np.zeros(ch, dtype=np.float32)
    lib().orc_tex3d_linear(_p(vol), ch, rx, ry, rz, u, v, w, _p(out))
    return out


def tex2d(img, u, v):
    img = f32(img)
    if img.ndim == 2:
        img = img[..., None]
    h, w, ch = img.shape
    out = np.zeros(ch, dtype=np.float32)
    lib().orc_tex2d_linear(_p(img), ch, w, h, u, v, _p(out))
    return out


def morph(depth, mode=0):
    depth = f32(depth)
    h, w = depth.shape
    out = np.empty_like(depth)
    lib().orc_morph(_p(depth), w, h, C.c_uint(mode), _p(out))
    return out


def rgb_to_lab(rgb):
    a = f32(rgb)
    out = np.zeros(3, dtype=np.float32)
    lib().orc_rgb_to_lab(_p(a), _p(out))
    return out


def decode_dxt(blocks, W, H, mode):
    b = np.ascontiguousarray(blocks, dtype=np.uint8)
    out = np.zeros((H, W, 3), dtype=np.uint8)
    lib().orc_decode_dxt(_p(b), W, H, mode, _p(out))
    return out


def pre_depth(depth, color, cv_xyz, cv_uv, limits, bbox_min, bbox_max, filter_textures=True, compress=False,
              near=0.5, far=4.5):
    depth, cv_xyz, cv_uv = f32(depth), f32(cv_xyz), f32(cv_uv)
    color = np.ascontiguousarray(color, dtype=np.uint8)
    h, w = depth.shape
    p = DepthParams()
    p.W, p.H = w, h
    p.Hc, p.Wc = color.shape[0], color.shape[1]
    p.xyz_res[:] = [cv_xyz.shape[2], cv_xyz.shape[1], cv_xyz.shape[0]]
    p.uv_res[:] = [cv_uv.shape[2], cv_uv.shape[1], cv_uv.shape[0]]
    p.cv_min_ds, p.cv_max_ds = limits
    p.bbox_min[:] = bbox_min
    p.bbox_max[:] = bbox_max
    p.filter_textures = int(filter_textures)
    p.compress = int(compress)
    p.near_, p.far_ = near, far
    out_rg = np.empty((h, w, 2), dtype=np.float32)
    out_lab = np.empty((h, w, 3), dtype=np.float32)
    lib().orc_pre_depth(_p(depth), _p(color), _p(cv_xyz), _p(cv_uv), C.byref(p), _p(out_rg), _p(out_lab))
    return out_rg, out_lab


def boundary(depth_rg, lab, refine=True):
    depth_rg, lab = f32(depth_rg), f32(lab)
    h, w = depth_rg.shape[:2]
    out_b = np.empty((h, w, 2), dtype=np.float32)
    out_s = np.empty((h, w), dtype=np.float32)
    lib().orc_boundary(_p(depth_rg), _p(lab), w, h, int(refine), _p(out_b), _p(out_s))
    return out_b, out_s


def normal(depth_b, cv_xyz, bbox_min, bbox_max, brick_size, res_bricks, bricks=None):
    depth_b, cv_xyz = f32(depth_b), f32(cv_xyz)
    h, w = depth_b.shape[:2]
    p = NormalParams()
    p.W, p.H = w, h
    p.xyz_res[:] = [cv_xyz.shape[2], cv_xyz.shape[1], cv_xyz.shape[0]]
    p.bbox_min[:] = bbox_min
    p.bbox_max[:] = bbox_max
    p.brick_size = brick_size
    p.res_bricks[:] = list(res_bricks)
    out = np.empty((h, w, 3), dtype=np.float32)
    lib().orc_normal(_p(depth_b), _p(cv_xyz), C.byref(p), _p(out), _p(bricks) if bricks is not None else None)
    return out


def quality(depth_b, normals, cv_xyz, cam_pos):
    depth_b, normals, cv_xyz = f32(depth_b), f32(normals), f32(cv_xyz)
    h, w = depth_b.shape[:2]
    res = (C.c_int * 3)(cv_xyz.shape[2], cv_xyz.shape[1], cv_xyz.shape[0])
    cam = f32(cam_pos)
    out = np.empty((h, w), dtype=np.float32)
    lib().orc_quality(_p(depth_b), _p(normals), _p(cv_xyz), res, _p(cam), w, h, _p(out))
    return out


def camera_pos(cv_xyz):
    cv_xyz = f32(cv_xyz)
    res = (C.c_int * 3)(cv_xyz.shape[2], cv_xyz.shape[1], cv_xyz.shape[0])
    out = np.zeros(3, dtype=np.float32)
    lib().orc_camera_pos(_p(cv_xyz), res, _p(out))
    return out


def brick_voxel_mask(bbox_min, bbox_max, brick_size, res, occupied_mask, z_range=None):
    """The voxels the reference draws in brick mode: divideBox + containedVoxels, literally.
    Returns (mask [z1-z0, Y, X] u8, res_bricks, number of listed index triples outside the volume)."""
    X, Y, Z = res
    z0, z1 = (0, Z) if z_range is None else z_range
    occ = np.ascontiguousarray(occupied_mask, dtype=np.uint8)
    rb = divide_box(bbox_min, bbox_max, brick_size)
    assert occ.size == rb[0] * rb[1] * rb[2], "occupied mask has %d bricks, divideBox makes %s" % (occ.size, rb)
    out = np.empty((z1 - z0, Y, X), dtype=np.uint8)
    d = (C.c_int * 3)(X, Y, Z)
    rbo = (C.c_int * 3)()
    outside = lib().orc_brick_voxel_mask(_p(f32(bbox_min)), _p(f32(bbox_max)), C.c_float(brick_size), d, _p(occ), z0, z1,
                                         _p(out), rbo)
    assert tuple(rbo) == rb
    return out, rb, int(outside)


def integrate(inv_luts, sils, depth_bs, quals, res, limit, occupied_mask=None, bv=None, res_bricks=None,
              z_range=None, out=None, bbox=None, brick_size=None):
    """inv_luts: list of [Iz,Iy,Ix,4]; images: lists of [H,W(,2)]; res = (X,Y,Z).
    Returns the full [Z,Y,X] volume (only z_range written when given).
    occupied_mask (one byte per brick, divideBox order) selects brick mode: the voxels drawn
    are those of brick_voxel_mask(bbox[0], bbox[1], brick_size, ...); `res_bricks`, if given
    (e.g. the product's geometry), must equal the oracle's own divideBox; `bv` is ignored."""
    n = len(inv_luts)
    inv = [f32(a) for a in inv_luts]
    sil = [f32(a) for a in sils]
    db = [f32(a) for a in depth_bs]
    q = [f32(a) for a in quals]
    h, w = sil[0].shape
    p = IntegrateParams()
    p.num_sensors, p.W, p.H = n, w, h
    p.res[:] = list(res)
    p.limit = limit
    arr = lambda xs: (C.c_void_p * n)(*[x.ctypes.data for x in xs])
    inv_res = (C.c_int * (3 * n))()
    for i, a in enumerate(inv):
        inv_res[3 * i:3 * i + 3] = [a.shape[2], a.shape[1], a.shape[0]]
    X, Y, Z = res
    if out is None:
        out = np.full((Z, Y, X), np.nan, dtype=np.float32)
    z0, z1 = (0, Z) if z_range is None else z_range
    vmask = None
    if occupied_mask is not None:
        assert bbox is not None and brick_size is not None, "brick mode needs bbox=(min, max) and brick_size"
        vmask, rb, _ = brick_voxel_mask(bbox[0], bbox[1], brick_size, res, occupied_mask, (z0, z1))
        if res_bricks is not None:
            assert tuple(res_bricks) == rb, "res_bricks %s differs from divideBox %s" % (tuple(res_bricks), rb)
    lib().orc_integrate(C.byref(p), arr(inv), inv_res, arr(sil), arr(db), arr(q),
                        _p(vmask) if vmask is not None else None, z0, z1, _p(out))
    return out


def frustum_planes(cv_xyz):
    cv_xyz = f32(cv_xyz)
    res = (C.c_int * 3)(cv_xyz.shape[2], cv_xyz.shape[1], cv_xyz.shape[0])
    out = np.zeros((6, 4), dtype=np.float32)
    lib().orc_frustum_planes(_p(cv_xyz), res, _p(out))
    return out


def inverse_volume(cv_xyz, bbox_min, bbox_max, vol_res, z_range=None):
    """CalibrationInverter::calculateInverseVolumes for one sensor: [Z,Y,X,4]"""
    cv_xyz = f32(cv_xyz)
    res = (C.c_int * 3)(cv_xyz.shape[2], cv_xyz.shape[1], cv_xyz.shape[0])
    vr = (C.c_int * 3)(*vol_res)
    z0, z1 = (0, vol_res[2]) if z_range is None else z_range
    out = np.empty((z1 - z0, vol_res[1], vol_res[0], 4), dtype=np.float32)
    lib().orc_inverse_volume(_p(cv_xyz), res, _p(f32(bbox_min)), _p(f32(bbox_max)), vr, z0, z1, _p(out))
    return out


class BrickGrid(C.Structure):
    _fields_ = [("bbox_min", C.c_float * 3), ("brick_size", C.c_float), ("res_bricks", C.c_int * 3)]


def depth_peels(view_bytes, bbox_min, brick_size, res_bricks, counters, mask):
    v = View.from_buffer_copy(view_bytes)
    g = BrickGrid()
    g.bbox_min[:] = list(bbox_min)
    g.brick_size = brick_size
    g.res_bricks[:] = list(res_bricks)
    c = np.ascontiguousarray(counters, dtype=np.uint32)
    m = np.ascontiguousarray(mask, dtype=np.uint8)
    out = np.empty((v.height, v.width, 4), dtype=np.float32)
    lib().orc_depth_peels(C.byref(v), C.byref(g), _p(c), _p(m), _p(out))
    return out


def raymarch(view_bytes, tsdf, inv_luts, uv_luts, colors, depth_bs, quals, limit=0.01, peels=None):
    """tsdf [Z,Y,X]; inv_luts [Iz,Iy,Ix,4]; uv_luts [Rz,Ry,Rx,2]; colors u8 [Hc,Wc,3]"""
    v = View.from_buffer_copy(view_bytes)
    n = len(inv_luts)
    tsdf = f32(tsdf)
    inv = [f32(a) for a in inv_luts]
    uv = [f32(a) for a in uv_luts]
    col = [np.ascontiguousarray(a, dtype=np.uint8) for a in colors]
    db = [f32(a) for a in depth_bs]
    q = [f32(a) for a in quals]
    p = RaymarchParams()
    p.num_sensors = n
    p.H, p.W = q[0].shape
    p.Hc, p.Wc = col[0].shape[:2]
    p.res[:] = [tsdf.shape[2], tsdf.shape[1], tsdf.shape[0]]
    p.limit = limit
    arr = lambda xs: (C.c_void_p * n)(*[x.ctypes.data for x in xs])
    inv_res = (C.c_int * (3 * n))()
    uv_res = (C.c_int * (3 * n))()
    for i in range(n):
        inv_res[3 * i:3 * i + 3] = [inv[i].shape[2], inv[i].shape[1], inv[i].shape[0]]
        uv_res[3 * i:3 * i + 3] = [uv[i].shape[2], uv[i].shape[1], uv[i].shape[0]]
    color = np.empty((v.height, v.width, 4), dtype=np.float32)
    depth = np.empty((v.height, v.width), dtype=np.float32)
    ns = np.empty((v.height, v.width), dtype=np.float32)
    pk = f32(peels) if peels is not None else None
    lib().orc_raymarch(C.byref(v), C.byref(p), _p(tsdf), arr(inv), inv_res, arr(uv), uv_res, arr(col), arr(db), arr(q),
                       _p(pk) if pk is not None else None, _p(color), _p(depth), _p(ns))
    return color, depth, ns


def fill_colors(color, depth, return_atlas=False):
    """-> (filled colour, depth) [, the native LOD atlas [H, 1.5 W, 4] after the inpaint pyramid]"""
    color, depth = f32(color), f32(depth)
    h, w = depth.shape
    oc = np.empty((h, w, 4), dtype=np.float32)
    od = np.empty((h, w), dtype=np.float32)
    atlas = np.empty((h, int(np.float32(w) * np.float32(1.5)), 4), dtype=np.float32) if return_atlas else None
    lib().orc_fill_colors(_p(color), _p(depth), w, h, _p(oc), _p(od), _p(atlas) if return_atlas else None)
    return (oc, od, atlas) if return_atlas else (oc, od)


def fill_layout(w, h):
    n, fw = C.c_int(), C.c_int()
    off = (C.c_int * 40)()
    res = (C.c_int * 40)()
    lib().orc_fill_layout(w, h, C.byref(n), C.byref(fw), off, res)
    return n.value, fw.value, np.array(off[:]).reshape(20, 2), np.array(res[:]).reshape(20, 2)


def volume_res(bbox_min, bbox_max, voxel):
    res = (C.c_int * 3)()
    lib().orc_volume_res(_p(f32(bbox_min)), _p(f32(bbox_max)), voxel, res)
    return tuple(res)


def divide_box(bbox_min, bbox_max, brick_size):
    res = (C.c_int * 3)()
    lib().orc_divide_box(_p(f32(bbox_min)), _p(f32(bbox_max)), brick_size, res)
    return tuple(res)


def adjust_brick_size(size, voxel):
    return float(lib().orc_adjust_brick_size(size, voxel))


def update_occupied(counters, min_voxels):
    c = np.ascontiguousarray(counters, dtype=np.uint32)
    ids = np.empty_like(c)
    ratio = C.c_float()
    n = lib().orc_update_occupied(_p(c), c.size, min_voxels, _p(ids), C.byref(ratio))
    return ids[:n].copy(), ratio.value


def reference_cpu_work(bbox_min, bbox_max, dims, brick_size, counters, min_voxels):
    """Times the reference's genuinely-CPU work on this path:
    VolumeSampler::resize, divideBox + containedVoxels, the updateOccupiedBricks filter."""
    import time

    l = lib()
    l.orc_contained_voxels.restype = C.c_double
    x, y, z = dims
    n = x * y * z
    pos = np.empty(n * 3, dtype=np.float32)
    t0 = time.perf_counter()
    l.orc_volume_sampler_resize(x, y, z, _p(pos))
    t_resize = time.perf_counter() - t0
    del pos
    idx = np.empty(n + n // 8, dtype=np.uint32)
    cnt = C.c_size_t()
    d = (C.c_int * 3)(x, y, z)
    t0 = time.perf_counter()
    l.orc_contained_voxels(_p(f32(bbox_min)), _p(f32(bbox_max)), C.c_float(brick_size), d, _p(idx), C.c_size_t(idx.size),
                           C.byref(cnt))
    t_div = time.perf_counter() - t0
    del idx
    c = np.ascontiguousarray(counters, dtype=np.uint32)
    ids = np.empty_like(c)
    ratio = C.c_float()
    t0 = time.perf_counter()
    for _ in range(10):
        l.orc_update_occupied(_p(c), c.size, min_voxels, _p(ids), C.byref(ratio))
    t_upd = (time.perf_counter() - t0) / 10
    return {"volume_sampler_resize_s": round(t_resize, 3), "divide_box_contained_voxels_s": round(t_div, 3),
            "brick_voxel_indices": int(cnt.value), "update_occupied_filter_ms": round(t_upd * 1e3, 4)}


def lut_write(path, data, floats):
    data = f32(data)
    rz, ry, rx = data.shape[:3]
    res = (C.c_uint32 * 3)(rx, ry, rz)
    lim = (C.c_float * 2)(0.5, 4.5)
    return lib().orc_lut_write(path.encode(), res, lim, _p(data), floats)


def lut_read(path, floats):
    res = (C.c_uint32 * 3)()
    lim = (C.c_float * 2)()
    rc = lib().orc_lut_read_header(path.encode(), res, lim)
    if rc != 0:
        raise IOError(path)
    out = np.empty((res[2], res[1], res[0], floats), dtype=np.float32)
    rc = lib().orc_lut_read(path.encode(), _p(out), floats)
    if rc != 0:
        raise IOError(path)
    return out, tuple(lim)


def run_pipeline(scene, bbox_min, bbox_max, res, inv_luts, limit=0.01, brick_size=None, bv=8, res_bricks=None,
                 min_voxels=10, filter_textures=True, processed=True, refine=True, use_bricks=True,
                 compress=False, depth_override=None, limits=(0.5, 4.5)):
    """Whole frame through the oracle, in the reference's call order
    (source/kinect_client.cpp:572-602).  Returns a dict of per-sensor images,
    brick counters, occupied ids and the TSDF volume.  `bv` is ignored (brick membership is
    the reference's containedVoxels, see brick_voxel_mask)."""
    n = scene.N
    # m_res_bricks comes from the oracle's own divideBox; a caller passing the product's geometry gets it checked
    rb = divide_box(bbox_min, bbox_max, brick_size)
    if res_bricks is not None:
        assert tuple(res_bricks) == rb, "res_bricks %s differs from divideBox %s" % (tuple(res_bricks), rb)
    res_bricks = rb
    nb = res_bricks[0] * res_bricks[1] * res_bricks[2]
    counters = np.zeros(nb, dtype=np.uint32)
    out = {k: [] for k in ("raw", "morph", "depth_rg", "lab", "depth_b", "sil", "normal", "quality")}
    depth_all = scene.depth if depth_override is None else depth_override
    for i in range(n):
        raw = f32(depth_all[i])
        m = morph(morph(raw, 0), 1)
        src = m if processed else raw
        rg, lab = pre_depth(src, scene.color[i], scene.xyz[i], scene.uv[i], limits, bbox_min, bbox_max,
                            filter_textures, compress)
        db, sil = boundary(rg, lab, refine)
        nrm = normal(db, scene.xyz[i], bbox_min, bbox_max, brick_size, res_bricks, counters)
        q = quality(db, nrm, scene.xyz[i], camera_pos(scene.xyz[i]))
        for k, v in zip(("raw", "morph", "depth_rg", "lab", "depth_b", "sil", "normal", "quality"),
                        (raw, m, rg, lab, db, sil, nrm, q)):
            out[k].append(v)
    ids, ratio = update_occupied(counters, min_voxels)
    mask = np.zeros(nb, dtype=np.uint8)
    mask[ids] = 1
    out["counters"], out["occupied"], out["ratio"], out["mask"] = counters, ids, ratio, mask
    if inv_luts is not None:
        out["tsdf"] = integrate(inv_luts, out["sil"], out["depth_b"], out["quality"], res, limit,
                                mask if use_bricks else None, bbox=(bbox_min, bbox_max), brick_size=brick_size)
    return out
