"""Deterministic synthetic scene of SURVEY.md section 8(d): N pinhole sensors on
a ring looking at a sphere standing over a floor plane, analytic calibration
volumes in the reference's formats (cv_xyz 12 B, cv_uv 8 B, cv_xyz_inv 16 B
records, x fastest -- framework/calibration/calibration_volume.hpp:57-59).

Pure numpy; used by tests and bench.py to make inputs.  It is neither the oracle
nor the product path.
"""
import numpy as np

BBOX_MIN = (-1.0, 0.0, -1.0)
BBOX_MAX = (1.0, 2.0, 1.0)
DEPTH_MIN, DEPTH_MAX = 0.5, 4.5


class Sensor:
    def __init__(self, index, num, width, height, radius=2.5, cam_height=1.2, centre=(0.0, 1.0, 0.0)):
        ang = 2.0 * np.pi * index / max(num, 1) + 0.3
        self.pos = np.array([radius * np.cos(ang), cam_height, radius * np.sin(ang)], dtype=np.float64)
        fwd = np.asarray(centre, dtype=np.float64) - self.pos
        self.forward = fwd / np.linalg.norm(fwd)
        r = np.cross(self.forward, np.array([0.0, 1.0, 0.0]))
        self.right = r / np.linalg.norm(r)
        # image y runs downwards (as in the reference's Kinect calibrations): the index
        # -> world map of cv_xyz is then left-handed, which is the orientation
        # Frustum::getPlanes (frustum.cpp:149-166) needs for its normals to point inwards
        self.up = -np.cross(self.right, self.forward)
        self.W, self.H = width, height
        self.fx = self.fy = 365.0 * width / 512.0
        self.cx, self.cy = width / 2.0, height / 2.0

    def rays(self, px, py):
        """un-normalised ray directions with unit z-depth for pixel coordinates"""
        x = (px - self.cx) / self.fx
        y = (py - self.cy) / self.fy
        return x[..., None] * self.right + y[..., None] * self.up + self.forward

    def project(self, world):
        d = world - self.pos
        xc, yc, zc = d @ self.right, d @ self.up, d @ self.forward
        with np.errstate(divide="ignore", invalid="ignore"):
            px = self.fx * xc / zc + self.cx
            py = self.fy * yc / zc + self.cy
        return px, py, zc


def render_depth(s, seed, noise_sigma=0.002, hole_fraction=0.02, sphere_c=(0.0, 1.0, 0.0), sphere_r=0.5,
                 floor_y=0.05):
    """z-depth image in metres: analytic ray cast + Gaussian noise + random holes."""
    py, px = np.meshgrid(np.arange(s.H) + 0.5, np.arange(s.W) + 0.5, indexing="ij")
    d = s.rays(px, py)
    o = s.pos
    t = np.full((s.H, s.W), np.inf)
    with np.errstate(divide="ignore", invalid="ignore"):
        tf = (floor_y - o[1]) / d[..., 1]
    tf = np.where((tf > 0) & np.isfinite(tf), tf, np.inf)
    hit = o + tf[..., None] * d
    inside = (np.abs(hit[..., 0]) < 1.5) & (np.abs(hit[..., 2]) < 1.5)
    t = np.minimum(t, np.where(inside, tf, np.inf))
    oc = o - np.asarray(sphere_c)
    a = np.sum(d * d, axis=-1)
    b = 2.0 * np.sum(d * oc, axis=-1)
    c = oc @ oc - sphere_r ** 2
    disc = b * b - 4 * a * c
    ts = np.where(disc > 0, (-b - np.sqrt(np.maximum(disc, 0))) / (2 * a), np.inf)
    ts = np.where(ts > 0, ts, np.inf)
    t = np.minimum(t, ts)
    rng = np.random.default_rng(seed)
    depth = np.where(np.isfinite(t), t, 0.0)
    depth = depth + rng.normal(0.0, noise_sigma, depth.shape) * (depth > 0)
    depth[rng.random(depth.shape) < hole_fraction] = 0.0
    depth[(depth < DEPTH_MIN) | (depth > DEPTH_MAX)] = 0.0
    return depth.astype(np.float32)


def render_shell_depth(s, seed, noise_sigma=0.002, shell_c=(0.0, 1.0, 0.0), shell_r=0.95):
    """The DENSE scene: the inside of a spherical shell inscribed in the box, seen through its (one-sided) near half.
    A sensor 1.4 m from the centre sees the shell over its whole field of view (asin(0.95 / 1.4) = 42.7 deg > the 42.3 deg
    half-diagonal of a 512 x 424 image at fx = 365), so EVERY pixel carries a measurement whose point lies inside the
    box: no holes, no background -- the worst case for the per-pixel work of the pre_* chain (169 taps everywhere,
    glsl/pre_depth.fs:85-127) and for anything that skips undecided or empty regions."""
    py, px = np.meshgrid(np.arange(s.H) + 0.5, np.arange(s.W) + 0.5, indexing="ij")
    d = s.rays(px, py)
    oc = s.pos - np.asarray(shell_c)
    a = np.sum(d * d, axis=-1)
    b = 2.0 * np.sum(d * oc, axis=-1)
    c = oc @ oc - shell_r ** 2
    disc = b * b - 4 * a * c
    t = np.where(disc > 0, (-b + np.sqrt(np.maximum(disc, 0))) / (2 * a), 0.0)     # the FAR intersection: the inner wall
    rng = np.random.default_rng(seed)
    depth = t + rng.normal(0.0, noise_sigma, t.shape) * (t > 0)
    depth[(depth < DEPTH_MIN) | (depth > DEPTH_MAX)] = 0.0
    return depth.astype(np.float32)


def render_color(s, seed, wh=None):
    """procedural RGB8 checker (optionally at a colour resolution != the depth resolution;
    cv_uv is normalised, so it addresses either)"""
    w, h = wh if wh else (s.W, s.H)
    y, x = np.meshgrid(np.arange(h) * s.H // h, np.arange(w) * s.W // w, indexing="ij")
    rng = np.random.default_rng(seed + 7919)
    base = ((x // 16 + y // 16) % 2).astype(np.float64)
    img = np.stack([40 + 180 * base, 60 + 120 * (1 - base), 90 + 100 * ((x // 32) % 2)], axis=-1)
    img += rng.integers(-8, 9, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


def forward_luts(s, res=(32, 27, 32)):
    """cv_xyz [Rz,Ry,Rx,3] and cv_uv [Rz,Ry,Rx,2]: texel (i,j,k) <-> pixel
    coordinate ((i+.5)/Rx*W, (j+.5)/Ry*H), depth dmin + (k+.5)/Rz*(dmax-dmin)."""
    rx, ry, rz = res
    u = (np.arange(rx) + 0.5) / rx
    v = (np.arange(ry) + 0.5) / ry
    w = (np.arange(rz) + 0.5) / rz
    W3, V3, U3 = np.meshgrid(w, v, u, indexing="ij")
    depth = DEPTH_MIN + W3 * (DEPTH_MAX - DEPTH_MIN)
    dirs = s.rays(U3 * s.W, V3 * s.H)
    xyz = s.pos + depth[..., None] * dirs
    uv = np.stack([U3, V3], axis=-1)
    return xyz.astype(np.float32), uv.astype(np.float32)


def inverse_lut(s, res, bbox_min=BBOX_MIN, bbox_max=BBOX_MAX, z_range=None):
    """cv_xyz_inv [Iz,Iy,Ix,4]: volume position -> (u, v, dnorm, 1) or -1 outside
    the frustum (calibration_inverter.cpp:127-141), sampled at cell centres."""
    ix, iy, iz = res
    z0, z1 = (0, iz) if z_range is None else z_range
    if (z1 - z0) * ix * iy > (1 << 21):      # slabs that stay in cache: 8x faster at 256^3, the same values bit for bit
        rows = max(1, (1 << 21) // (ix * iy))
        return np.concatenate([inverse_lut(s, res, bbox_min, bbox_max, (z, min(z + rows, z1))) for z in range(z0, z1, rows)])
    bmin, bmax = np.asarray(bbox_min, dtype=np.float64), np.asarray(bbox_max, dtype=np.float64)
    x = bmin[0] + (np.arange(ix) + 0.5) / ix * (bmax[0] - bmin[0])
    y = bmin[1] + (np.arange(iy) + 0.5) / iy * (bmax[1] - bmin[1])
    z = bmin[2] + (np.arange(z0, z1) + 0.5) / iz * (bmax[2] - bmin[2])
    Z3, Y3, X3 = np.meshgrid(z, y, x, indexing="ij")
    world = np.stack([X3, Y3, Z3], axis=-1)
    px, py, zc = s.project(world)
    u, v = px / s.W, py / s.H
    dn = (zc - DEPTH_MIN) / (DEPTH_MAX - DEPTH_MIN)
    ok = (zc >= DEPTH_MIN) & (zc <= DEPTH_MAX) & (u >= 0) & (u <= 1) & (v >= 0) & (v <= 1)
    out = np.full(world.shape[:-1] + (4,), -1.0, dtype=np.float32)
    out[..., 0] = np.where(ok, u, -1.0)
    out[..., 1] = np.where(ok, v, -1.0)
    out[..., 2] = np.where(ok, dn, -1.0)
    out[..., 3] = np.where(ok, 1.0, -1.0)
    return out


class Scene:
    """All inputs of one configuration."""

    def __init__(self, num_sensors, width, height, lut_res=(32, 27, 32), seed=1234, make_frames=True, sphere_r=None,
                 color_wh=None, layout="ring"):
        """layout "ring": SURVEY 8(d)'s scene (sensors 2.5 m out, a sphere over a floor; two thirds of the pixels see
        nothing).  layout "dense": sensors 1.4 m from the centre at its height, looking into a spherical shell that fills
        every field of view (render_shell_depth): every pixel valid and inside the box."""
        self.N, self.W, self.H = num_sensors, width, height
        self.layout, self.seed, self.color_wh = layout, seed, color_wh
        # SURVEY 8(d) scene: sphere r = 0.5 m.  At small test resolutions a 13x13
        # window spans most of such a sphere and the bilateral pass rejects every
        # pixel, so low-resolution scenes use a larger sphere to stay non-trivial.
        if sphere_r is None:
            sphere_r = 0.5 if width >= 256 else 0.9
        self.sphere_r = sphere_r
        self.lut_res = tuple(lut_res)
        if layout == "dense":
            self.sensors = [Sensor(i, num_sensors, width, height, radius=1.4, cam_height=1.0) for i in range(num_sensors)]
        elif layout == "ring":
            self.sensors = [Sensor(i, num_sensors, width, height) for i in range(num_sensors)]
        else:
            raise ValueError("layout is 'ring' or 'dense'")
        self.xyz, self.uv = [], []
        for s in self.sensors:
            a, b = forward_luts(s, lut_res)
            self.xyz.append(a)
            self.uv.append(b)
        if make_frames:
            self.depth, self.color = self.frame(0)

    def frame(self, k):
        """(depth [N,H,W] f32, colour [N,h,w,3] u8) of frame k of a MOVING sequence: frame 0 is the static scene; in frame
        k > 0 every sensor draws new noise and new holes, the ring scene's sphere has moved by k * (6, 0, 4) cm and the
        dense scene's shell by k * (2, 0, 1.5) cm (it stays inside the box) -- so occupied bricks, tile states, list
        sizes and store elision see change from one step to the next."""
        seed = self.seed + 1000 * k
        if self.layout == "dense":
            c = (0.02 * k, 1.0, 0.015 * k)
            depth = [render_shell_depth(s, seed + i, shell_c=c) for i, s in enumerate(self.sensors)]
        else:
            c = (0.06 * k, 1.0, 0.04 * k)
            depth = [render_depth(s, seed + i, sphere_c=c, sphere_r=self.sphere_r) for i, s in enumerate(self.sensors)]
        color = [render_color(s, seed + i, self.color_wh) for i, s in enumerate(self.sensors)]
        return np.stack(depth), np.stack(color)

    def at_frame(self, k):
        """a shallow copy of the scene holding frame k (for the oracle, which reads scene.depth / scene.color)"""
        import copy
        other = copy.copy(self)
        other.depth, other.color = self.frame(k)
        return other

    def inverse(self, res, bbox_min=BBOX_MIN, bbox_max=BBOX_MAX):
        return [inverse_lut(s, res, bbox_min, bbox_max) for s in self.sensors]

    def pinhole(self, i):
        from . import capi

        s = self.sensors[i]
        p = capi.Pinhole()
        p.cam_pos[:] = s.pos.tolist()
        p.right[:] = s.right.tolist()
        p.up[:] = s.up.tolist()
        p.forward[:] = s.forward.tolist()
        p.fx, p.fy, p.cx, p.cy = s.fx, s.fy, s.cx, s.cy
        p.depth_min, p.depth_max = DEPTH_MIN, DEPTH_MAX
        p.lut_res[:] = list(self.lut_res)
        return p


def compress_depth_u8(depth_m, near=0.5, far=4.5):
    """inverse of pre_depth.fs uncompress() (sqrt mapping), for u8 depth tests"""
    scale = far - near
    sn = scale / 255.0
    v = np.sqrt(np.maximum((depth_m - near) / scale - 0.15 * sn, 0.0))
    out = np.where(depth_m > 0, np.clip(np.round(v * 255.0), 0, 255), 0)
    return out.astype(np.uint8)


def encode_dxt(img, mode=1):
    """Simple DXT1 / DXT5 encoder for test inputs (per 4x4 block: endpoints = max /
    min colour in RGB565, nearest of the four palette entries).  Returns the block
    stream in the layout squish / GL use (blocks row-major, 8 or 16 bytes each)."""
    H, W = img.shape[:2]
    bh, bw = (H + 3) // 4, (W + 3) // 4
    pad = np.zeros((bh * 4, bw * 4, 3), np.int32)
    pad[:H, :W] = img
    pad[H:, :W] = pad[H - 1:H, :W]
    pad[:, W:] = pad[:, W - 1:W]
    blk = pad.reshape(bh, 4, bw, 4, 3).transpose(0, 2, 1, 3, 4).reshape(bh * bw, 16, 3)
    hi, lo = blk.max(axis=1), blk.min(axis=1)

    def to565(c):
        return ((c[:, 0] >> 3) << 11) | ((c[:, 1] >> 2) << 5) | (c[:, 2] >> 3)

    def from565(v):
        r, g, b = (v >> 11) & 31, (v >> 5) & 63, v & 31
        return np.stack([(r << 3) | (r >> 2), (g << 2) | (g >> 4), (b << 3) | (b >> 2)], axis=-1)

    c0, c1 = to565(hi), to565(lo)
    swap = c0 < c1
    c0, c1 = np.where(swap, c1, c0), np.where(swap, c0, c1)            # c0 >= c1
    e0, e1 = from565(c0), from565(c1)
    four = (c0 > c1) | (mode != 1)                                      # 4-colour mode
    p2 = np.where(four[:, None], (2 * e0 + e1) // 3, (e0 + e1) // 2)
    p3 = np.where(four[:, None], (e0 + 2 * e1) // 3, 0)
    pal = np.stack([e0, e1, p2, p3], axis=1)                            # [nb, 4, 3]
    dist = ((blk[:, :, None, :] - pal[:, None, :, :]) ** 2).sum(-1)    # [nb, 16, 4]
    if mode == 1:
        dist[:, :, 3] += np.where(four, 0, 1 << 30)[:, None]           # never pick transparent black
    idx = dist.argmin(-1).astype(np.uint32)                            # [nb, 16]
    bits = (idx << (2 * np.arange(16, dtype=np.uint32))[None, :]).sum(axis=1).astype(np.uint32)
    out = np.zeros((bh * bw, 8 if mode == 1 else 16), np.uint8)
    o = 0 if mode == 1 else 8
    out[:, o + 0], out[:, o + 1] = c0 & 255, c0 >> 8
    out[:, o + 2], out[:, o + 3] = c1 & 255, c1 >> 8
    for k in range(4):
        out[:, o + 4 + k] = (bits >> (8 * k)) & 255
    if mode != 1:
        out[:, 0] = out[:, 1] = 255                                      # opaque alpha block
    return out.reshape(-1)
