"""ctypes bindings of include/rgbdr.h (one-to-one; see that header for the
reference interface each entry point replaces)."""
import ctypes as C
import os

import numpy as np

MAX_SENSORS = 8
TILE = 8

OK = 0
ERR_INVALID_ARGUMENT = -1
ERR_OUT_OF_RANGE = -2
ERR_NO_DEVICE = -3
ERR_HIP = -4
ERR_IO = -5
ERR_STATE = -6
ERR_NO_MEMORY = -7

FLAG_FILTER, FLAG_PROCESSED, FLAG_REFINE, FLAG_USE_BRICKS = 1, 2, 4, 8
FLAGS_DEFAULT = 15
FLAG_PIPELINE = 16
FLAG_NO_RESAMPLE = 32
FLAG_ELIDE_STORES = 64
FLAG_SKIP_BACKGROUND = 128

IMG_DEPTH_RAW, IMG_DEPTH_MORPH, IMG_DEPTH_RG, IMG_LAB, IMG_DEPTH_B_RG, IMG_SILHOUETTE, IMG_NORMAL, IMG_QUALITY = range(8)
IMG_COLOR = 8
IMG_CHANNELS = {IMG_DEPTH_RAW: 1, IMG_DEPTH_MORPH: 1, IMG_DEPTH_RG: 2, IMG_LAB: 3, IMG_DEPTH_B_RG: 2,
                IMG_SILHOUETTE: 1, IMG_NORMAL: 3, IMG_QUALITY: 1}


class Config(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("num_sensors", C.c_int32),
        ("depth_w", C.c_int32), ("depth_h", C.c_int32),
        ("color_w", C.c_int32), ("color_h", C.c_int32),
        ("bbox_min", C.c_float * 3), ("bbox_max", C.c_float * 3),
        ("voxel_size", C.c_float), ("brick_size", C.c_float), ("tsdf_limit", C.c_float),
        ("min_voxels_per_brick", C.c_uint32),
        ("flags", C.c_uint32),
        ("compress_depth", C.c_int32), ("compress_rgb", C.c_int32),
        ("near_", C.c_float * MAX_SENSORS), ("far_", C.c_float * MAX_SENSORS),
        ("res_override", C.c_int32 * 3),
        ("slab_rank", C.c_int32), ("slab_count", C.c_int32),
    ]


class Lut(C.Structure):
    _fields_ = [("res", C.c_uint32 * 3), ("depth_limits", C.c_float * 2), ("data", C.c_void_p)]


class Geometry(C.Structure):
    _fields_ = [
        ("res_volume", C.c_int32 * 3), ("res_bricks", C.c_int32 * 3),
        ("brick_size", C.c_float), ("brick_voxels", C.c_int32), ("brick_voxels_axis", C.c_int32 * 3),
        ("num_bricks", C.c_int32),
        ("tiles", C.c_int32 * 3),
        ("slab_tile_z0", C.c_int32), ("slab_tile_z1", C.c_int32),
        ("slab_voxel_z0", C.c_int32), ("slab_voxel_z1", C.c_int32),
        ("halo_tile_layers", C.c_int32),
    ]


class Pinhole(C.Structure):
    _fields_ = [
        ("cam_pos", C.c_float * 3),
        ("right", C.c_float * 3), ("up", C.c_float * 3), ("forward", C.c_float * 3),
        ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
        ("depth_min", C.c_float), ("depth_max", C.c_float),
        ("lut_res", C.c_int32 * 3),
    ]


class View(C.Structure):
    """rgbdr_view: the uniforms of tsdf_raymarch.{vs,fs}"""
    _fields_ = [("modelview", C.c_float * 16), ("projection", C.c_float * 16), ("normal_matrix", C.c_float * 16),
                ("gl_normal_matrix_inv", C.c_float * 16), ("vol_to_world", C.c_float * 16),
                ("vol_to_world_inv", C.c_float * 16), ("modelview_inv", C.c_float * 16), ("img_to_eye", C.c_float * 16),
                ("camera_pos", C.c_float * 3), ("width", C.c_int32), ("height", C.c_int32), ("shade_mode", C.c_int32),
                ("skip_space", C.c_int32)]


HALO_PEER_BYTES = 1024     # sizeof(rgbdr_halo_peer)


def make_view(eye, target, up, fovy_deg, width, height, bbox_min, bbox_max, near=0.1, far=10.0, shade_mode=0):
    """Builds the ray-marcher's uniforms the way ReconIntegration::draw does
    (recon_integration.cpp:177-241) from a look-at camera and a perspective
    projection; float64 algebra, stored as float32 column-major."""
    eye, target, up = (np.asarray(a, dtype=np.float64) for a in (eye, target, up))
    f = target - eye
    f /= np.linalg.norm(f)
    s = np.cross(f, up)
    s /= np.linalg.norm(s)
    u = np.cross(s, f)
    mv = np.eye(4)
    mv[0, :3], mv[1, :3], mv[2, :3] = s, u, -f
    mv[:3, 3] = -mv[:3, :3] @ eye
    t = 1.0 / np.tan(np.radians(fovy_deg) / 2.0)
    proj = np.zeros((4, 4))
    proj[0, 0], proj[1, 1] = t / (width / height), t
    proj[2, 2], proj[2, 3], proj[3, 2] = (far + near) / (near - far), 2 * far * near / (near - far), -1.0
    bmin, bmax = np.asarray(bbox_min, dtype=np.float64), np.asarray(bbox_max, dtype=np.float64)
    v2w = np.eye(4)
    v2w[:3, :3] = np.diag(bmax - bmin)
    v2w[:3, 3] = bmin
    vp_scale = np.diag([width * 0.5, height * 0.5, 0.5, 1.0])
    vp_trans = np.eye(4)
    vp_trans[:3, 3] = 1.0
    v = View()

    def put(name, m):
        getattr(v, name)[:] = np.asarray(m, dtype=np.float32).T.reshape(-1).tolist()   # column-major

    put("modelview", mv)
    put("projection", proj)
    put("normal_matrix", np.linalg.inv(mv @ v2w).T)
    put("gl_normal_matrix_inv", np.linalg.inv(np.linalg.inv(mv).T))
    put("vol_to_world", v2w)
    put("vol_to_world_inv", np.linalg.inv(v2w))
    put("modelview_inv", np.linalg.inv(mv))
    put("img_to_eye", np.linalg.inv(vp_scale @ vp_trans @ proj))
    cam = np.linalg.inv(v2w) @ (np.linalg.inv(mv) @ np.array([0.0, 0.0, 0.0, 1.0]))
    v.camera_pos[:] = cam[:3].astype(np.float32).tolist()
    v.width, v.height, v.shade_mode, v.skip_space = width, height, shade_mode, 0
    return v


class TsdfDeviceView(C.Structure):
    _fields_ = [("base", C.c_void_p), ("owned", C.c_void_p), ("layer_bytes", C.c_size_t),
                ("owned_layers", C.c_int32), ("halo_layers", C.c_int32)]


class ImageDeviceView(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32), ("channels", C.c_int32),
                ("element_bytes", C.c_int32), ("stream", C.c_void_p)]


class CalibrationDeviceView(C.Structure):
    _fields_ = [("cv_xyz", C.c_void_p), ("cv_uv", C.c_void_p), ("xyz_res", C.c_uint32 * 3), ("uv_res", C.c_uint32 * 3),
                ("inv_res", C.c_uint32 * 3), ("depth_limits", C.c_float * 2), ("stream", C.c_void_p)]


class ShardDeviceView(C.Structure):
    _fields_ = [("frames", C.c_void_p), ("sensor_bytes", C.c_size_t), ("num_sensors", C.c_int32), ("first", C.c_int32),
                ("count", C.c_int32), ("counters", C.c_void_p), ("num_bricks", C.c_uint32), ("stream", C.c_void_p)]


# every symbol include/rgbdr.h declares: name -> (restype, argtypes)
_P = C.c_void_p
_CFG, _GEO, _LUT = C.POINTER(Config), C.POINTER(Geometry), C.POINTER(Lut)
_F, _U32 = C.POINTER(C.c_float), C.POINTER(C.c_uint32)
SYMBOLS = {
    "rgbdr_create": (C.c_int, [_CFG, C.c_int, C.POINTER(_P)]),
    "rgbdr_destroy": (None, [_P]),
    "rgbdr_last_error": (C.c_char_p, [_P]),
    "rgbdr_status_string": (C.c_char_p, [C.c_int]),
    "rgbdr_version": (C.c_char_p, []),
    "rgbdr_compute_geometry": (C.c_int, [_CFG, _GEO]),
    "rgbdr_brick_voxel_range": (C.c_int, [_CFG, C.c_int, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "rgbdr_slab_range": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "rgbdr_camera_position": (C.c_int, [_LUT, _F]),
    "rgbdr_set_calibration": (C.c_int, [_P, C.c_int, _LUT, _LUT]),
    "rgbdr_set_inverse_calibration": (C.c_int, [_P, C.c_int, _LUT]),
    "rgbdr_load_calibration_files": (C.c_int, [_P, C.c_int, C.c_char_p, C.c_char_p, C.c_char_p]),
    "rgbdr_synth_inverse_calibration": (C.c_int, [_P, C.c_int, C.POINTER(Pinhole)]),
    "rgbdr_compute_inverse_calibration": (C.c_int, [_P, C.c_int, C.c_int]),
    "rgbdr_generate_inverse_lut": (C.c_int, [_P, C.c_int, _U32, C.c_int, _F]),
    "rgbdr_inverse_search_stats": (C.c_int, [_P, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "rgbdr_upload_frame": (C.c_int, [_P, _P, _P]),
    "rgbdr_upload_frame_device": (C.c_int, [_P, _P, _P]),
    "rgbdr_clear_occupied_bricks": (C.c_int, [_P]),
    "rgbdr_process_textures": (C.c_int, [_P]),
    "rgbdr_update_occupied_bricks": (C.c_int, [_P]),
    "rgbdr_set_occupied_bricks": (C.c_int, [_P, _U32, C.c_size_t]),
    "rgbdr_integrate": (C.c_int, [_P]),
    "rgbdr_step": (C.c_int, [_P, _P, _P]),
    "rgbdr_sync": (C.c_int, [_P]),
    "rgbdr_set_voxel_size": (C.c_int, [_P, C.c_float]),
    "rgbdr_set_tsdf_limit": (C.c_int, [_P, C.c_float]),
    "rgbdr_set_brick_size": (C.c_int, [_P, C.c_float]),
    "rgbdr_set_use_bricks": (C.c_int, [_P, C.c_int]),
    "rgbdr_set_pipelined": (C.c_int, [_P, C.c_int]),
    "rgbdr_set_elide_stores": (C.c_int, [_P, C.c_int]),
    "rgbdr_set_skip_background": (C.c_int, [_P, C.c_int]),
    "rgbdr_set_sweep_launches": (C.c_int, [_P, C.c_int]),
    "rgbdr_skipped_pairs": (C.c_int, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "rgbdr_readback_skip_tables": (C.c_int, [_P, C.c_int, _P, C.c_size_t]),
    "rgbdr_set_min_voxels_per_brick": (C.c_int, [_P, C.c_uint32]),
    "rgbdr_filter_textures": (C.c_int, [_P, C.c_int]),
    "rgbdr_use_processed_depths": (C.c_int, [_P, C.c_int]),
    "rgbdr_refine_boundary": (C.c_int, [_P, C.c_int]),
    "rgbdr_get_brick_size": (C.c_float, [_P]),
    "rgbdr_occupied_ratio": (C.c_float, [_P]),
    "rgbdr_num_bricks": (C.c_uint32, [_P]),
    "rgbdr_get_geometry": (C.c_int, [_P, _GEO]),
    "rgbdr_get_camera_position": (C.c_int, [_P, C.c_int, _F]),
    "rgbdr_readback_tsdf": (C.c_int, [_P, _F]),
    "rgbdr_readback_image": (C.c_int, [_P, C.c_int, C.c_int, _F]),
    "rgbdr_readback_inverse_calibration": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _F]),
    "rgbdr_readback_color": (C.c_int, [_P, C.c_int, C.POINTER(C.c_uint8)]),
    "rgbdr_readback_brick_counters": (C.c_int, [_P, _U32]),
    "rgbdr_get_occupied": (C.c_int, [_P, _U32, C.c_size_t, C.POINTER(C.c_size_t), _F]),
    "rgbdr_device_tsdf": (C.c_int, [_P, C.POINTER(TsdfDeviceView)]),
    "rgbdr_device_frame": (C.c_int, [_P, C.c_int, C.POINTER(_P)]),
    "rgbdr_device_image": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(ImageDeviceView)]),
    "rgbdr_device_calibration": (C.c_int, [_P, C.c_int, C.POINTER(CalibrationDeviceView)]),
    "rgbdr_raymarch": (C.c_int, [_P, C.POINTER(View), _F, _F, _F]),
    "rgbdr_fill_colors": (C.c_int, [_P, _F, _F]),
    "rgbdr_draw": (C.c_int, [_P, _P, C.c_int]),
    "rgbdr_device_view_frame": (C.c_int, [_P, C.c_int, C.POINTER(_P), C.POINTER(_P), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "rgbdr_device_view_frame_async": (C.c_int, [_P, C.c_int, C.POINTER(_P), C.POINTER(_P), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(_P)]),
    "rgbdr_readback_view_frame": (C.c_int, [_P, C.c_int, _F, _F]),
    "rgbdr_map_frame_buffer": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_P), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "rgbdr_upload_mapped_frame": (C.c_int, [_P]),
    "rgbdr_halo_staging": (C.c_int, [_P, C.c_int, C.POINTER(_P), C.POINTER(_P), C.POINTER(C.c_size_t)]),
    "rgbdr_set_halo_staging": (C.c_int, [_P, C.c_int]),
    "rgbdr_readback_tile_layers": (C.c_int, [_P, C.c_int, C.c_int, _F]),
    "rgbdr_halo_exchange": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P]),
    "rgbdr_halo_begin_step": (C.c_int, [_P]),
    "rgbdr_halo_exchange_async": (C.c_int, [_P, _P, C.c_int, C.c_int]),
    "rgbdr_halo_export": (C.c_int, [_P, _P]),
    "rgbdr_halo_set_peer": (C.c_int, [_P, C.c_int, _P]),
    "rgbdr_halo_pull_async": (C.c_int, [_P]),
    "rgbdr_halo_wait": (C.c_int, [_P]),
    "rgbdr_set_sensor_shard": (C.c_int, [_P, C.c_int, C.c_int]),
    "rgbdr_shard_view": (C.c_int, [_P, C.POINTER(ShardDeviceView)]),
    "rgbdr_shard_allgather": (C.c_int, [_P, _P]),
    "rgbdr_shard_gather_done": (C.c_int, [_P]),
    "rgbdr_import_frame": (C.c_int, [_P, _P, _P, _P]),
    "rgbdr_shard_allgather_async": (C.c_int, [_P, _P]),
    "rgbdr_import_frame_from": (C.c_int, [_P, _P]),
    "rgbdr_settle": (C.c_int, [_P, C.c_float, C.POINTER(C.c_float)]),
    "rgbdr_get_arena_probe": (C.c_int, [_P, _F, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "rgbdr_get_arena_chunks": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_float)]),
    "rgbdr_upload_view_frame": (C.c_int, [_P, C.c_int, C.c_int, _F, _F]),
    "rgbdr_raymarch_find": (C.c_int, [_P, C.POINTER(View), C.POINTER(_P)]),
    "rgbdr_raymarch_shade": (C.c_int, [_P, C.POINTER(View), _F, _F, _F]),
    "rgbdr_draw_depth_limits": (C.c_int, [_P, C.POINTER(View), _F]),
    "rgbdr_stream": (_P, [_P]),
    "rgbdr_set_stream": (C.c_int, [_P, _P]),
    "rgbdr_enable_timers": (C.c_int, [_P, C.c_int]),
    "rgbdr_timer_ns": (C.c_int, [_P, C.c_char_p, C.POINTER(C.c_uint64)]),
    "rgbdr_enable_timer_accumulation": (C.c_int, [_P, C.c_int]),
    "rgbdr_timer_stats": (C.c_int, [_P, C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]),
    "rgbdr_set_timer_detail": (C.c_int, [_P, C.c_int]),
}

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "librgbdr_hip.so")
_lib = None


class RgbdrError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("rgbdr status %d: %s" % (status, message))
        self.status = status


def lib():
    """Load librgbdr_hip.so; raises (never falls back) when it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing: run __graft_entry__.build() (make -C rgbd-recon_amd/csrc)" % LIB_PATH)
        # The PyTorch wheel bundles its own HIP runtime.  If this library pulled in
        # /opt/rocm's libamdhip64 first, a later torch.cuda initialisation in the same
        # process would find two runtimes and report "No HIP GPUs".  Importing torch
        # first makes its runtime the one both sides share (harness concern only:
        # a C/C++ host links librgbdr_hip.so against the system ROCm directly).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def make_config(num_sensors, depth_wh, color_wh=None, bbox_min=(-1.0, 0.0, -1.0), bbox_max=(1.0, 2.0, 1.0),
                voxel_size=0.01, brick_size=None, tsdf_limit=0.01, min_voxels=10, flags=FLAGS_DEFAULT,
                compress_depth=0, compress_rgb=0, near=0.5, far=4.5, res_override=(0, 0, 0), slab_rank=0,
                slab_count=1):
    c = Config()
    c.struct_size = C.sizeof(Config)
    c.num_sensors = num_sensors
    c.depth_w, c.depth_h = depth_wh
    c.color_w, c.color_h = color_wh if color_wh else depth_wh
    c.bbox_min[:] = bbox_min
    c.bbox_max[:] = bbox_max
    c.voxel_size = voxel_size
    c.brick_size = brick_size if brick_size is not None else 8 * voxel_size
    c.tsdf_limit = tsdf_limit
    c.min_voxels_per_brick = min_voxels
    c.flags = flags
    c.compress_depth = compress_depth
    c.compress_rgb = compress_rgb
    for i in range(MAX_SENSORS):
        c.near_[i] = near
        c.far_[i] = far
    c.res_override[:] = res_override
    c.slab_rank, c.slab_count = slab_rank, slab_count
    return c


def make_lut(data, res, limits=(0.5, 4.5)):
    """rgbdr_lut over a contiguous float32 array; keeps the array alive."""
    arr = np.ascontiguousarray(data, dtype=np.float32)
    l = Lut()
    l.res[:] = [int(r) for r in res]
    l.depth_limits[:] = [float(limits[0]), float(limits[1])]
    l.data = arr.ctypes.data
    l._keep = arr
    return l


def compute_geometry(cfg):
    g = Geometry()
    rc = lib().rgbdr_compute_geometry(C.byref(cfg), C.byref(g))
    if rc != OK:
        raise RgbdrError(rc, lib().rgbdr_last_error(None).decode())
    return g


def camera_position(cv_xyz, res):
    """rgbdr_camera_position: Frustum(getCornerPoints(cv_xyz)).getCameraPos() of a forward calibration volume (no context needed)"""
    out = (C.c_float * 3)()
    lut = make_lut(cv_xyz, res)
    rc = lib().rgbdr_camera_position(C.byref(lut), out)
    if rc != OK:
        raise RgbdrError(rc, lib().rgbdr_last_error(None).decode())
    return np.array(out[:], dtype=np.float32)


def brick_voxel_range(cfg, axis, brick):
    """(first, last) voxel index of `brick` on `axis` as the reference's containedVoxels builds it"""
    a, b = C.c_int32(), C.c_int32()
    rc = lib().rgbdr_brick_voxel_range(C.byref(cfg), axis, brick, C.byref(a), C.byref(b))
    if rc != OK:
        raise RgbdrError(rc, lib().rgbdr_last_error(None).decode())
    return a.value, b.value


class Context:
    """Thin RAII wrapper; method names follow the reference's
    (NetKinectArray::update/processTextures, ReconIntegration::integrate ...)."""

    def __init__(self, cfg, device=0):
        self._h = _P()
        self.cfg = cfg
        rc = lib().rgbdr_create(C.byref(cfg), device, C.byref(self._h))
        if rc != OK:
            raise RgbdrError(rc, lib().rgbdr_last_error(None).decode())
        self.geo = self.geometry()

    def _chk(self, rc):
        if rc != OK:
            raise RgbdrError(rc, lib().rgbdr_last_error(self._h).decode())

    def close(self):
        if self._h:
            lib().rgbdr_destroy(self._h)
            self._h = _P()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def geometry(self):
        g = Geometry()
        self._chk(lib().rgbdr_get_geometry(self._h, C.byref(g)))
        return g

    # calibration
    def set_calibration(self, sensor, xyz, xyz_res, uv, uv_res, limits):
        a, b = make_lut(xyz, xyz_res, limits), make_lut(uv, uv_res, limits)
        self._chk(lib().rgbdr_set_calibration(self._h, sensor, C.byref(a), C.byref(b)))

    def set_inverse_calibration(self, sensor, inv, res):
        a = make_lut(inv, res, (0.5, 4.5))
        self._chk(lib().rgbdr_set_inverse_calibration(self._h, sensor, C.byref(a)))

    def load_calibration_files(self, sensor, xyz=None, uv=None, inv=None):
        enc = lambda s: s.encode() if s else None
        self._chk(lib().rgbdr_load_calibration_files(self._h, sensor, enc(xyz), enc(uv), enc(inv)))

    def compute_inverse_calibration(self, sensor, window=0):
        self._chk(lib().rgbdr_compute_inverse_calibration(self._h, sensor, window))

    def generate_inverse_lut(self, sensor, res, window=0):
        out = np.empty((res[2], res[1], res[0], 4), dtype=np.float32)
        r = (C.c_uint32 * 3)(*res)
        self._chk(lib().rgbdr_generate_inverse_lut(self._h, sensor, r, window, out.ctypes.data_as(_F)))
        return out

    def inverse_search_stats(self, sensor):
        """(voxels whose window was widened, voxels searched exhaustively) in the sensor's last inverse-LUT search"""
        w, e = C.c_uint64(0), C.c_uint64(0)
        self._chk(lib().rgbdr_inverse_search_stats(self._h, sensor, C.byref(w), C.byref(e)))
        return int(w.value), int(e.value)

    def synth_inverse_calibration(self, sensor, pinhole):
        self._chk(lib().rgbdr_synth_inverse_calibration(self._h, sensor, C.byref(pinhole)))

    def camera_position(self, sensor):
        out = (C.c_float * 3)()
        self._chk(lib().rgbdr_get_camera_position(self._h, sensor, out))
        return np.array(out[:], dtype=np.float32)

    # per frame
    def update(self, depth, color):
        d = np.ascontiguousarray(depth)
        c = np.ascontiguousarray(color, dtype=np.uint8)   # RGB8 pixels or DXT blocks
        self._chk(lib().rgbdr_upload_frame(self._h, d.ctypes.data, c.ctypes.data))

    def map_frame_buffer(self):
        """(depth, color) numpy uint8 views of the page-locked back buffer of the double frame buffer"""
        d, c = _P(), _P()
        nd, nc = C.c_size_t(), C.c_size_t()
        self._chk(lib().rgbdr_map_frame_buffer(self._h, C.byref(d), C.byref(c), C.byref(nd), C.byref(nc)))
        depth = np.ctypeslib.as_array((C.c_uint8 * nd.value).from_address(d.value))
        color = np.ctypeslib.as_array((C.c_uint8 * nc.value).from_address(c.value))
        return depth, color

    def upload_mapped_frame(self):
        self._chk(lib().rgbdr_upload_mapped_frame(self._h))

    def update_device(self, depth_ptr, color_ptr):
        self._chk(lib().rgbdr_upload_frame_device(self._h, depth_ptr, color_ptr))

    def clear_occupied_bricks(self):
        self._chk(lib().rgbdr_clear_occupied_bricks(self._h))

    def process_textures(self):
        self._chk(lib().rgbdr_process_textures(self._h))

    def update_occupied_bricks(self):
        self._chk(lib().rgbdr_update_occupied_bricks(self._h))

    def set_occupied_bricks(self, ids):
        a = np.ascontiguousarray(ids, dtype=np.uint32)
        self._chk(lib().rgbdr_set_occupied_bricks(self._h, a.ctypes.data_as(_U32), a.size))

    def integrate(self):
        self._chk(lib().rgbdr_integrate(self._h))

    def step(self, depth, color):
        d = np.ascontiguousarray(depth)
        c = np.ascontiguousarray(color, dtype=np.uint8)
        self._chk(lib().rgbdr_step(self._h, d.ctypes.data, c.ctypes.data))

    def sync(self):
        self._chk(lib().rgbdr_sync(self._h))

    # setters
    def set_voxel_size(self, v):
        self._chk(lib().rgbdr_set_voxel_size(self._h, v))
        self.cfg.voxel_size = v
        self.cfg.res_override[:] = [0, 0, 0]
        self.geo = self.geometry()

    def set_brick_size(self, v):
        self._chk(lib().rgbdr_set_brick_size(self._h, v))
        self.cfg.brick_size = v
        self.geo = self.geometry()

    def set_tsdf_limit(self, v):
        self._chk(lib().rgbdr_set_tsdf_limit(self._h, v))
        self.cfg.tsdf_limit = v

    def _flag(self, flag, on):
        self.cfg.flags = (self.cfg.flags | flag) if on else (self.cfg.flags & ~flag)

    def set_use_bricks(self, on):
        self._chk(lib().rgbdr_set_use_bricks(self._h, int(on)))
        self._flag(FLAG_USE_BRICKS, on)

    def set_pipelined(self, on):
        self._chk(lib().rgbdr_set_pipelined(self._h, int(on)))
        self._flag(FLAG_PIPELINE, on)

    def set_elide_stores(self, on):
        self._chk(lib().rgbdr_set_elide_stores(self._h, int(on)))
        self._flag(FLAG_ELIDE_STORES, on)

    def set_sweep_launches(self, n):
        """the full sweep as n launches (a kernel waiting on another queue gets onto the device between two)"""
        self._chk(lib().rgbdr_set_sweep_launches(self._h, int(n)))

    def set_skip_background(self, on):
        self._chk(lib().rgbdr_set_skip_background(self._h, int(on)))
        self._flag(FLAG_SKIP_BACKGROUND, on)

    def skipped_pairs(self):
        """(skipped, total) (tile, sensor) pairs of a RGBDR_FLAG_SKIP_BACKGROUND sweep of the current frame"""
        a, b = C.c_uint64(), C.c_uint64()
        self._chk(lib().rgbdr_skipped_pairs(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def readback_skip_tables(self, which):
        """diagnostic tables of RGBDR_FLAG_SKIP_BACKGROUND: 0 verdict bytes [tiles][N]; 1 (origins int32, dmin f32,
        dmax f32, size class int32) each [tiles][N]; 2 bounds f32 [N][size class][3][H+1][W+1]"""
        n = self.cfg.num_sensors
        pairs = self.skipped_pairs()[1]
        if which == 0:
            out = np.empty((pairs // n, n), dtype=np.uint8)
        elif which == 1:
            out = np.empty((4, pairs // n, n), dtype=np.int32)
        else:
            out = np.empty((n, 3, 3, self.cfg.depth_h + 1, self.cfg.depth_w + 1), dtype=np.float32)
        self._chk(lib().rgbdr_readback_skip_tables(self._h, which, out.ctypes.data, out.nbytes))
        if which == 1:
            return out[0], out[1].view(np.float32), out[2].view(np.float32), out[3]
        return out

    def set_min_voxels_per_brick(self, n):
        self._chk(lib().rgbdr_set_min_voxels_per_brick(self._h, n))
        self.cfg.min_voxels_per_brick = n

    def filter_textures(self, on):
        self._chk(lib().rgbdr_filter_textures(self._h, int(on)))
        self._flag(FLAG_FILTER, on)

    def use_processed_depths(self, on):
        self._chk(lib().rgbdr_use_processed_depths(self._h, int(on)))
        self._flag(FLAG_PROCESSED, on)

    def refine_boundary(self, on):
        self._chk(lib().rgbdr_refine_boundary(self._h, int(on)))
        self._flag(FLAG_REFINE, on)

    def occupied_ratio(self):
        return float(lib().rgbdr_occupied_ratio(self._h))

    # outputs
    def readback_tsdf(self):
        g = self.geo
        out = np.empty((g.slab_voxel_z1 - g.slab_voxel_z0, g.res_volume[1], g.res_volume[0]), dtype=np.float32)
        self._chk(lib().rgbdr_readback_tsdf(self._h, out.ctypes.data_as(_F)))
        return out

    def readback_image(self, which, sensor):
        ch = IMG_CHANNELS.get(which, 1)
        out = np.empty((self.cfg.depth_h, self.cfg.depth_w, ch), dtype=np.float32)
        self._chk(lib().rgbdr_readback_image(self._h, which, sensor, out.ctypes.data_as(_F)))
        return out[..., 0] if ch == 1 else out

    def readback_inverse_calibration(self, sensor, z0, z1, res_xy=None):
        x, y = res_xy if res_xy else (self.geo.res_volume[0], self.geo.res_volume[1])
        out = np.empty((z1 - z0, y, x, 4), dtype=np.float32)
        self._chk(lib().rgbdr_readback_inverse_calibration(self._h, sensor, z0, z1, out.ctypes.data_as(_F)))
        return out

    def readback_color(self, sensor):
        out = np.empty((self.cfg.color_h, self.cfg.color_w, 3), dtype=np.uint8)
        self._chk(lib().rgbdr_readback_color(self._h, sensor, out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out

    def readback_brick_counters(self):
        out = np.empty(self.geo.num_bricks, dtype=np.uint32)
        self._chk(lib().rgbdr_readback_brick_counters(self._h, out.ctypes.data_as(_U32)))
        return out

    def get_occupied(self):
        ids = np.empty(self.geo.num_bricks, dtype=np.uint32)
        n = C.c_size_t()
        ratio = C.c_float()
        self._chk(lib().rgbdr_get_occupied(self._h, ids.ctypes.data_as(_U32), ids.size, C.byref(n), C.byref(ratio)))
        return ids[: n.value].copy(), ratio.value

    def raymarch(self, view):
        """(color [H,W,4], depth [H,W], num_samples [H,W]) of ReconIntegration::draw"""
        h, w = view.height, view.width
        color = np.empty((h, w, 4), dtype=np.float32)
        depth = np.empty((h, w), dtype=np.float32)
        ns = np.empty((h, w), dtype=np.float32)
        self._chk(lib().rgbdr_raymarch(self._h, C.byref(view), color.ctypes.data_as(_F), depth.ctypes.data_as(_F),
                                       ns.ctypes.data_as(_F)))
        return color, depth, ns

    def raymarch_find(self, view):
        """device pointer (int) of the height*width int32 first-hit buffer of this slab"""
        ptr = _P()
        self._chk(lib().rgbdr_raymarch_find(self._h, C.byref(view), C.byref(ptr)))
        return ptr.value

    def raymarch_shade(self, view):
        h, w = view.height, view.width
        color = np.empty((h, w, 4), dtype=np.float32)
        depth = np.empty((h, w), dtype=np.float32)
        ns = np.empty((h, w), dtype=np.float32)
        self._chk(lib().rgbdr_raymarch_shade(self._h, C.byref(view), color.ctypes.data_as(_F), depth.ctypes.data_as(_F),
                                             ns.ctypes.data_as(_F)))
        return color, depth, ns

    def draw_depth_limits(self, view):
        out = np.empty((view.height, view.width, 4), dtype=np.float32)
        self._chk(lib().rgbdr_draw_depth_limits(self._h, C.byref(view), out.ctypes.data_as(_F)))
        return out

    def fill_colors(self, width, height):
        """ReconIntegration::fillColors on the last ray-marched frame"""
        color = np.empty((height, width, 4), dtype=np.float32)
        depth = np.empty((height, width), dtype=np.float32)
        self._chk(lib().rgbdr_fill_colors(self._h, color.ctypes.data_as(_F), depth.ctypes.data_as(_F)))
        return color, depth

    def draw(self, view, fill_holes=True):
        """ReconIntegration::drawF: depth peels (view.skip_space), ray-march, hole filling -- enqueued, not waited for"""
        self._chk(lib().rgbdr_draw(self._h, C.byref(view), 1 if fill_holes else 0))

    def device_view_frame(self, filled):
        """(colour pointer, depth pointer, width, height) of the displayed frame on the device"""
        c, d, w, h = _P(), _P(), C.c_int(), C.c_int()
        self._chk(lib().rgbdr_device_view_frame(self._h, 1 if filled else 0, C.byref(c), C.byref(d), C.byref(w), C.byref(h)))
        return c.value, d.value, w.value, h.value

    def device_view_frame_async(self, filled):
        """the same without making the context's stream wait: (colour pointer, depth pointer, width, height, hipEvent_t) -- the
        caller's own queue waits for the event before it reads"""
        c, d, w, h, e = _P(), _P(), C.c_int(), C.c_int(), _P()
        self._chk(lib().rgbdr_device_view_frame_async(self._h, 1 if filled else 0, C.byref(c), C.byref(d), C.byref(w), C.byref(h), C.byref(e)))
        return c.value, d.value, w.value, h.value, e.value

    def readback_view_frame(self, filled):
        _, _, w, h = self.device_view_frame(filled)
        color = np.empty((h, w, 4), dtype=np.float32)
        depth = np.empty((h, w), dtype=np.float32)
        self._chk(lib().rgbdr_readback_view_frame(self._h, 1 if filled else 0, color.ctypes.data_as(_F), depth.ctypes.data_as(_F)))
        return color, depth

    # the halo by copy engine (rgbdr_halo_export / _set_peer / _pull_async): exports are plain bytes
    def halo_exchange(self, nccl_comm, peer_lo, peer_hi, buffer=-1, hip_stream=None):
        """one grouped send / recv of the boundary layers on a stream (rgbdr_halo_exchange)"""
        self._chk(lib().rgbdr_halo_exchange(self._h, nccl_comm, peer_lo, peer_hi, buffer, _P(hip_stream) if hip_stream else None))

    def halo_export(self):
        buf = C.create_string_buffer(HALO_PEER_BYTES)
        self._chk(lib().rgbdr_halo_export(self._h, buf))
        return buf.raw

    def halo_set_peer(self, side, export_bytes):
        if export_bytes is None:
            self._chk(lib().rgbdr_halo_set_peer(self._h, side, None))
            return
        assert len(export_bytes) == HALO_PEER_BYTES
        self._chk(lib().rgbdr_halo_set_peer(self._h, side, C.create_string_buffer(export_bytes, HALO_PEER_BYTES)))

    def halo_pull_async(self):
        self._chk(lib().rgbdr_halo_pull_async(self._h))

    def halo_staging(self, buffer):
        """(lo_ptr, hi_ptr, bytes) of halo staging set `buffer` (0 / 1)"""
        lo, hi, n = _P(), _P(), C.c_size_t()
        self._chk(lib().rgbdr_halo_staging(self._h, buffer, C.byref(lo), C.byref(hi), C.byref(n)))
        return lo.value, hi.value, n.value

    def set_halo_staging(self, buffer):
        self._chk(lib().rgbdr_set_halo_staging(self._h, buffer))

    def settle(self, max_seconds=4.0):
        """wait until the device streams steadily (background wipe of released memory); ms of the last replay"""
        ms = C.c_float()
        self._chk(lib().rgbdr_settle(self._h, max_seconds, C.byref(ms)))
        return float(ms.value)

    def arena_probe(self):
        """([ms per candidate placement of the LUT arena], index kept)"""
        ms = (C.c_float * 16)()
        n, chosen = C.c_int(), C.c_int()
        self._chk(lib().rgbdr_get_arena_probe(self._h, ms, C.byref(n), C.byref(chosen)))
        return [round(float(ms[i]), 4) for i in range(n.value)], chosen.value

    def arena_chunks(self):
        """(chunks, ms): the LUT arena as a range of the fastest physical chunks (0 chunks: a plain allocation)"""
        n, ms = C.c_int(), C.c_float()
        self._chk(lib().rgbdr_get_arena_chunks(self._h, C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def upload_view_frame(self, color, depth):
        color = np.ascontiguousarray(color, dtype=np.float32)
        depth = np.ascontiguousarray(depth, dtype=np.float32)
        h, w = depth.shape
        assert color.shape == (h, w, 4)
        self._chk(lib().rgbdr_upload_view_frame(self._h, w, h, color.ctypes.data_as(_F), depth.ctypes.data_as(_F)))

    # the managed halo exchange of the C ABI (side stream, staging sets and events owned by the context; RCCL bound at run
    # time): nccl_comm is a raw ncclComm_t (rgbd_recon_amd.dist.RcclComm.handle)
    def halo_begin_step(self):
        self._chk(lib().rgbdr_halo_begin_step(self._h))

    def halo_exchange_async(self, nccl_comm, peer_lo, peer_hi):
        self._chk(lib().rgbdr_halo_exchange_async(self._h, nccl_comm, peer_lo, peer_hi))

    def halo_wait(self):
        self._chk(lib().rgbdr_halo_wait(self._h))

    def set_sensor_shard(self, first, count):
        """process_textures runs the pre_* chain for sensors [first, first + count) only (0, 0: all again)"""
        self._chk(lib().rgbdr_set_sensor_shard(self._h, first, count))

    def shard_view(self):
        v = ShardDeviceView()
        self._chk(lib().rgbdr_shard_view(self._h, C.byref(v)))
        return v

    def shard_gather_done(self):
        """the host has enqueued its own collectives on shard_view().stream: the frame is complete in stream order"""
        self._chk(lib().rgbdr_shard_gather_done(self._h))

    def import_frame(self, frames_ptr, counters_ptr=None, wait_event=None):
        """packed frames (and brick counters) of another context's chain, in device memory; wait_event: a hipEvent_t handle"""
        self._chk(lib().rgbdr_import_frame(self._h, _P(int(frames_ptr)), _P(int(counters_ptr)) if counters_ptr else None,
                                           _P(int(wait_event)) if wait_event else None))

    def shard_allgather_async(self, nccl_comm):
        self._chk(lib().rgbdr_shard_allgather_async(self._h, nccl_comm))

    def import_frame_from(self, producer):
        self._chk(lib().rgbdr_import_frame_from(self._h, producer._h))

    def shard_allgather(self, nccl_comm):
        self._chk(lib().rgbdr_shard_allgather(self._h, nccl_comm))

    def device_image(self, which, sensor):
        v = ImageDeviceView()
        self._chk(lib().rgbdr_device_image(self._h, which, sensor, C.byref(v)))
        return v

    def device_calibration(self, sensor):
        """CalibVolumes::getXYZVolumeUnits / getUVVolumeUnits / getVolumeRes / getDepthLimits of one sensor, zero-copy"""
        v = CalibrationDeviceView()
        self._chk(lib().rgbdr_device_calibration(self._h, sensor, C.byref(v)))
        return v

    def device_tsdf(self):
        v = TsdfDeviceView()
        self._chk(lib().rgbdr_device_tsdf(self._h, C.byref(v)))
        return v

    def stream(self):
        return lib().rgbdr_stream(self._h)

    def set_stream(self, hip_stream):
        """hipStream_t handle as an int (e.g. torch.cuda.Stream.cuda_stream); None = own stream"""
        self._chk(lib().rgbdr_set_stream(self._h, _P(hip_stream) if hip_stream else None))

    def enable_timers(self, on=True):
        self._chk(lib().rgbdr_enable_timers(self._h, int(on)))

    def enable_timer_accumulation(self, on=True):
        self._chk(lib().rgbdr_enable_timer_accumulation(self._h, int(on)))

    def set_timer_detail(self, detail):
        self._chk(lib().rgbdr_set_timer_detail(self._h, detail))

    def timer_stats(self, name):
        """(total_ns, count) over the intervals since the last call; resets them"""
        ns, n = C.c_uint64(), C.c_uint32()
        self._chk(lib().rgbdr_timer_stats(self._h, name.encode(), C.byref(ns), C.byref(n)))
        return ns.value, n.value

    def timer_ns(self, name):
        ns = C.c_uint64()
        self._chk(lib().rgbdr_timer_ns(self._h, name.encode(), C.byref(ns)))
        return ns.value
