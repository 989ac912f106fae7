"""Known-answer tests of the oracle's restatements of the consumers of the volume and of the offline inverter
(SURVEY.md section 8 f-2, f-3, f-4): glsl/tsdf_raymarch.fs, glsl/shading.glsl, glsl/bricks.{vs,gs,fs} +
ReconIntegration::drawDepthLimits, CalibrationInverter::calculateInverseVolumes.  Every expected value below is
worked out by hand from the shader / C++ text (float64 arithmetic on the analytic scene), never taken from the
oracle; tests/mutation_check.py shows that these tests notice a flipped detail in the oracle.

Scene of the ray-march cases: the volume is the unit cube at the origin (vol_to_world = identity), the camera sits
on the axis x = y = 0.5 and looks along -z, the viewport has an odd size so that the centre pixel's ray is the axis.
The TSDF is a plane, tsdf(z) = clamp(zs - z, -limit, +limit) (negative towards the camera, positive behind the
surface), on a grid fine enough in z that LINEAR sampling is exact inside the band."""
import math

import numpy as np

LIMIT = 0.01
SD = LIMIT * 0.5             # sampleDistance = limit * 0.5f  (tsdf_raymarch.fs:33)
NEAR, FAR = 0.1, 10.0
W = H = 8                    # sensor images
C = 2                        # centre pixel of the 5 x 5 viewport


def plane_tsdf(zs, res=(4, 4, 512)):
    X, Y, Z = res
    zc = (np.arange(Z) + 0.5) / Z
    t = np.clip(zs - zc, -LIMIT, LIMIT).astype(np.float32)
    return np.broadcast_to(t[:, None, None], (Z, Y, X)).copy()


def identity_inverse(G=8):
    c = (np.arange(G) + 0.5) / G
    Z, Y, X = np.meshgrid(c, c, c, indexing="ij")
    return np.stack([X, Y, Z, np.ones_like(X)], axis=-1).astype(np.float32)


def uv_lut(G=4):
    c = (np.arange(G) + 0.5) / G
    Z, Y, X = np.meshgrid(c, c, c, indexing="ij")
    return np.stack([X, Y], axis=-1).astype(np.float32)


def sensors(depths, quals, cols):
    """per sensor: identity inverse LUT (pos_calib = sample_pos), constant depth / quality images, a constant colour"""
    n = len(depths)
    inv = [identity_inverse() for _ in range(n)]
    uv = [uv_lut() for _ in range(n)]
    col = [np.broadcast_to(np.array(c, np.uint8), (4, 4, 3)).copy() for c in cols]
    db = [np.stack([np.full((H, W), d, np.float32), np.zeros((H, W), np.float32)], axis=-1) for d in depths]
    q = [np.full((H, W), x, np.float32) for x in quals]
    return inv, uv, col, db, q


def view(pkg, eye_z=3.0, fov=20.0, n=5, mode=0, skip=0):
    v = pkg.capi.make_view((0.5, 0.5, eye_z), (0.5, 0.5, 0.5), (0, 1, 0), fov, n, n, (0, 0, 0), (1, 1, 1), near=NEAR, far=FAR,
                           shade_mode=mode)
    v.skip_space = skip
    return v


def window_depth(z, eye_z=3.0):
    """gl_FragDepth of a point on the axis at volume z (tsdf_raymarch.fs:133 with the perspective matrix of
    ReconIntegration::draw): (P[2][2] * vz + P[3][2]) / -vz * 0.5 + 0.5"""
    vz = -(eye_z - z)
    p22, p32 = (FAR + NEAR) / (NEAR - FAR), 2 * FAR * NEAR / (NEAR - FAR)
    return (p22 * vz + p32) / -vz * 0.5 + 0.5


ZS = 0.6123
DEFAULT_SENSORS = ([ZS + 0.002, ZS - 0.004, ZS + 0.02], [0.5, 1.0, 1.0], [(255, 0, 0), (0, 0, 255), (0, 255, 0)])


def march(orc, pkg, tsdf, sens=DEFAULT_SENSORS, peels=None, **kw):
    v = view(pkg, **kw)
    return orc.raymarch(bytes(v), tsdf, *sensors(*sens), limit=LIMIT, peels=peels)


# ---- tsdf_raymarch.fs: main() -------------------------------------------------------------------------------
def test_raymarch_hits_the_plane_where_the_secant_puts_it(orc, pkg):
    """samples sit at z_k = 1 - k * limit/2 (the ray enters the cube at z = 1, :76-84); the first with density > 0 is
    k = floor((1 - zs) / sd) + 1; with a TSDF that is linear between the two samples the refinement
    (sample_pos - step) - step * prev / (density - prev)  (:101) lands on zs itself"""
    color, depth, ns = march(orc, pkg, plane_tsdf(ZS))
    k_hit = math.floor((1.0 - ZS) / SD) + 1
    assert k_hit == 78
    assert ns[C, C] == np.float32(np.float32(k_hit + 1) * np.float32(0.0027))       # writeNumSamples(num_samples), :404-407
    assert abs(depth[C, C] - window_depth(ZS)) < 2e-7                                  # half a step off would be 4e-5
    assert color[C, C, 3] == 1.0


def test_raymarch_without_a_surface_discards_after_every_sample(orc, pkg):
    """max_num_samples = ceil(|t_far - t_near|) with t in units of sampleStep: the cube is 1 / sd = 200 steps deep;
    nothing found -> writeNumSamples(200), discard (the cleared colour (0,1,0,0) and depth 1 stay, :111-112)"""
    color, depth, ns = march(orc, pkg, np.full((512, 4, 4), -LIMIT, np.float32))
    assert ns[C, C] == np.float32(np.float32(200) * np.float32(0.0027))
    assert color[C, C].tolist() == [0.0, 1.0, 0.0, 0.0] and depth[C, C] == 1.0
    # density == 0 is not inside the contour (density > IsoValue, :99)
    color, depth, ns = march(orc, pkg, np.zeros((512, 4, 4), np.float32))
    assert ns[C, C] == np.float32(np.float32(200) * np.float32(0.0027)) and depth[C, C] == 1.0


def test_raymarch_first_sample_inside_uses_the_assumed_outside_density(orc, pkg):
    """prev_density starts at -limit (:91): a volume that is +limit everywhere is hit by the first sample at the cube
    face z = 1, and the secant moves back (1 + prev / (density - prev)) = half a step: z = 1 + sd / 2"""
    _, depth, ns = march(orc, pkg, np.full((512, 4, 4), LIMIT, np.float32))
    assert ns[C, C] == np.float32(0.0027)
    assert abs(depth[C, C] - window_depth(1.0 + SD / 2)) < 2e-7


def test_raymarch_camera_inside_the_cube_starts_at_the_camera(orc, pkg):
    """t_near < 0 -> 0 (:80): samples at z_k = 0.9 - k * sd"""
    _, depth, ns = march(orc, pkg, plane_tsdf(ZS), eye_z=0.9)
    k_hit = math.floor((0.9 - ZS) / SD) + 1
    assert k_hit == 58 and ns[C, C] == np.float32(np.float32(k_hit + 1) * np.float32(0.0027))
    assert abs(depth[C, C] - window_depth(ZS, eye_z=0.9)) < 2e-6


def test_raymarch_pixels_whose_ray_misses_the_cube_stay_cleared(orc, pkg):
    """the unit cube is not rasterised there: no fragment, so not even the sample-count image is written"""
    color, depth, ns = march(orc, pkg, plane_tsdf(ZS), fov=90.0)
    assert ns[0, 0] == 0.0 and color[0, 0].tolist() == [0.0, 1.0, 0.0, 0.0] and depth[0, 0] == 1.0
    assert ns[C, C] > 0.0 and depth[C, C] < 1.0


# ---- submitFragment / get_gradient / shading.glsl -------------------------------------------------------------
def test_raymarch_gradient_normal_points_out_of_the_surface(orc, pkg):
    """get_gradient = -normalize(central differences) (:144-157): the density grows along -z, so the normal is +z,
    towards the camera; shade mode 2 returns inverse(gl_NormalMatrix) * view_normal = the world normal
    (shading.glsl:62-63), and the modelview of this camera is a pure translation"""
    color, _, _ = march(orc, pkg, plane_tsdf(ZS), mode=2)
    assert np.allclose(color[C, C, :3], [0.0, 0.0, 1.0], atol=1e-6) and color[C, C, 3] == 1.0


def test_raymarch_blends_colours_by_quality_over_distance(orc, pkg):
    """blendColors (:303-338): weight = quality / (|depth - pos_calib.z| + 0.01), quality only when the distance is
    below limit; sensor 0: red, q 0.5, 2 mm off; sensor 1: blue, q 1, 4 mm off; sensor 2: green, 2 cm off -> no weight"""
    color, _, _ = march(orc, pkg, plane_tsdf(ZS))
    w0, w1 = 0.5 / (0.002 + 0.01), 1.0 / (0.004 + 0.01)
    assert np.allclose(color[C, C], [w0 / (w0 + w1), 0.0, w1 / (w0 + w1), 1.0], atol=2e-5)


def test_raymarch_fallback_blend_when_no_sensor_is_in_range(orc, pkg):
    """total_weight == 0 -> colours weighted by 1 / distance, alpha -1 (:331-336)"""
    sens = ([ZS + 0.02, ZS - 0.04], [1.0, 1.0], [(255, 0, 0), (0, 0, 255)])
    color, _, _ = march(orc, pkg, plane_tsdf(ZS), sens=sens)
    w0, w1 = 1.0 / 0.02, 1.0 / 0.04
    assert np.allclose(color[C, C], [w0 / (w0 + w1), 0.0, w1 / (w0 + w1), -1.0], atol=2e-5)


def test_raymarch_camera_influence_mode(orc, pkg):
    """shade mode 3 = blendCameras (:354-369): camera_colors[i] (shading.glsl:23-29) weighted by getWeights (quality
    where the distance is below limit); white when nothing weighs"""
    color, _, _ = march(orc, pkg, plane_tsdf(ZS), mode=3)
    cam = np.array([[228, 26, 28], [55, 126, 184]], np.float64) / 255.0
    exp = (cam[0] * 0.5 + cam[1] * 1.0) / 1.5
    assert np.allclose(color[C, C, :3], exp, atol=1e-6) and color[C, C, 3] == 1.0
    sens = ([ZS + 0.02], [1.0], [(255, 0, 0)])
    color, _, _ = march(orc, pkg, plane_tsdf(ZS), sens=sens, mode=3)
    assert color[C, C].tolist() == [1.0, 1.0, 1.0, 1.0]


def test_raymarch_phong_mode(orc, pkg):
    """shade mode 1 (shading.glsl:32-61) at view position (0, 0, -(3 - zs)) with view normal (0, 0, 1): light at
    (1.5, 1, 1) in view space, ambient 0.2 * diffuse * 0.5, diffuse (1, 0.9, 0.7) * 0.5 * cos, specular 0.5 * cos_h^20
    faded by 1 - (1 - cos)^6; alpha is the blend's"""
    color, _, _ = march(orc, pkg, plane_tsdf(ZS), mode=1)
    pos = np.array([0.0, 0.0, -(3.0 - ZS)])
    nrm = np.array([0.0, 0.0, 1.0])
    to_light = np.array([1.5, 1.0, 1.0]) - pos
    to_light /= np.linalg.norm(to_light)
    la = float(nrm @ to_light)
    half = to_light + (-pos / np.linalg.norm(pos))
    half /= np.linalg.norm(half)
    spec = float(half @ nrm) ** 20 * (1.0 - ((1.0 - la) ** 2) ** 3)
    ld = np.array([1.0, 0.9, 0.7])
    exp = ld * 0.2 * 0.5 + ld * 0.5 * la + 1.0 * 0.5 * spec
    assert la > 0 and np.allclose(color[C, C, :3], exp, atol=2e-6) and color[C, C, 3] == 1.0


def test_raymarch_surface_beyond_the_far_plane_is_dropped(orc, pkg):
    """gl_FragDepth is clamped to the depth range and tested GL_LESS against the cleared 1.0"""
    v = pkg.capi.make_view((0.5, 0.5, 3.0), (0.5, 0.5, 0.5), (0, 1, 0), 20.0, 5, 5, (0, 0, 0), (1, 1, 1), near=0.1, far=2.2)
    color, depth, ns = orc.raymarch(bytes(v), plane_tsdf(ZS), *sensors(*DEFAULT_SENSORS), limit=LIMIT)
    assert ns[C, C] > 0 and depth[C, C] == 1.0 and color[C, C].tolist() == [0.0, 1.0, 0.0, 0.0]      # 3 - zs = 2.39 > far


# ---- getStartPos (:392-401) -------------------------------------------------------------------------------------
def peel_image(r, g, b, n=5):
    p = np.zeros((n, n, 4), np.float32)
    p[..., 0], p[..., 1], p[..., 2] = r, g, b
    return p


def test_start_position_from_the_depth_peels(orc, pkg):
    """front face at z = 0.8, back face at z = 0.4: the march starts at the front face and takes
    ceil(distance / sd) = 80 samples at most; the plane is hit at k = floor((0.8 - zs) / sd) + 1"""
    peels = peel_image(window_depth(0.8), -window_depth(0.4), 1.0)
    _, depth, ns = march(orc, pkg, plane_tsdf(ZS), peels=peels, skip=1)
    k_hit = math.floor((0.8 - ZS) / SD) + 1
    assert k_hit == 38 and ns[C, C] == np.float32(np.float32(k_hit + 1) * np.float32(0.0027))
    assert abs(depth[C, C] - window_depth(ZS)) < 2e-6
    # the surface lies behind the back face: every sample of the interval is taken, nothing is found
    _, depth, ns = march(orc, pkg, plane_tsdf(0.3), peels=peels, skip=1)
    assert depth[C, C] == 1.0 and round(float(ns[C, C]) / 0.0027) in (80, 81)


def test_start_position_when_the_front_face_is_culled(orc, pkg):
    """closest face is a back face (r >= b): the front face was clipped, start at gl_DepthRange.near = the near plane,
    0.1 in front of the camera (z = 2.9); the samples outside the cube read the clamped edge texels"""
    d_back = window_depth(0.4)
    peels = peel_image(d_back, -d_back, d_back)
    _, depth, ns = march(orc, pkg, plane_tsdf(ZS), peels=peels, skip=1)
    k_hit = math.floor((2.9 - ZS) / SD) + 1
    assert abs(float(ns[C, C]) / 0.0027 - (k_hit + 1)) <= 1 and abs(depth[C, C] - window_depth(ZS)) < 2e-6


def test_start_position_without_any_brick_on_the_ray(orc, pkg):
    """cleared peel (1, 0, 1): r >= 1 -> pos_back = pos_front, zero samples, discard"""
    _, depth, ns = march(orc, pkg, plane_tsdf(ZS), peels=peel_image(1.0, 0.0, 1.0), skip=1)
    assert ns[C, C] == 0.0 and depth[C, C] == 1.0
    # (the cleared texel takes the r >= b branch first and ends with both points on the near plane; the shader's own
    # "no valid closest face" test, r >= 1.0 -> pos_back = pos_front (:398-399), needs b above 1 to be reached)
    _, depth, ns = march(orc, pkg, plane_tsdf(ZS), peels=peel_image(1.0, -window_depth(0.4), 1.5), skip=1)
    assert ns[C, C] == 0.0 and depth[C, C] == 1.0


# ---- bricks.{vs,gs,fs} + drawDepthLimits (recon_integration.cpp:409-429) --------------------------------------
GRID = ((0.0, 0.0, 0.0), 0.2, (5, 5, 5))         # bricks of 0.2: brick (2, 2, k) spans x, y in [0.4, 0.6], z in [0.2 k, 0.2 k + 0.2]


def peels_of(orc, pkg, occupied, counters=None):
    rb = GRID[2]
    mask = np.zeros(rb[0] * rb[1] * rb[2], np.uint8)
    cnt = np.zeros(rb[0] * rb[1] * rb[2], np.uint32)
    for (x, y, z) in occupied:
        mask[(z * rb[1] + y) * rb[0] + x] = 1
        cnt[(z * rb[1] + y) * rb[0] + x] = 10
    for (x, y, z), c in (counters or {}).items():
        cnt[(z * rb[1] + y) * rb[0] + x] = c
    return orc.depth_peels(bytes(view(pkg)), GRID[0], GRID[1], rb, cnt, mask)


def test_depth_peels_of_one_brick(orc, pkg):
    """MIN blending of (z, -z, front ? 1 : z) over the cleared (1, 0, 1, 0): r = nearest face, -g = farthest face,
    b = nearest back face; the camera looks along -z, so the brick's z = 0.8 face is its front"""
    p = peels_of(orc, pkg, [(2, 2, 3)])
    zf, zb = window_depth(0.8), window_depth(0.6)
    assert abs(p[C, C, 0] - zf) < 2e-6 and abs(p[C, C, 1] + zb) < 2e-6 and abs(p[C, C, 2] - zb) < 2e-6 and p[C, C, 3] == 0.0
    assert zf < zb
    assert p[0, 0].tolist() == [1.0, 0.0, 1.0, 0.0]               # a ray that meets no occupied brick: the cleared texel
    rb = GRID[2]
    mask = np.zeros(rb[0] * rb[1] * rb[2], np.uint8)
    mask[(3 * rb[1] + 2) * rb[0] + 2] = 1
    wide = orc.depth_peels(bytes(view(pkg, fov=120.0)), GRID[0], GRID[1], rb, mask.astype(np.uint32) * 10, mask)
    assert wide[0, 0].tolist() == [1.0, 0.0, 1.0, 0.0]            # ... and one that misses the brick grid altogether


def test_depth_peels_of_two_bricks_on_the_ray(orc, pkg):
    p = peels_of(orc, pkg, [(2, 2, 3), (2, 2, 1)])
    assert abs(p[C, C, 0] - window_depth(0.8)) < 2e-6               # nearest front
    assert abs(p[C, C, 1] + window_depth(0.2)) < 2e-6               # farthest back
    assert abs(p[C, C, 2] - window_depth(0.6)) < 2e-6               # nearest back face


def test_depth_peels_cull_faces_towards_a_full_neighbour(orc, pkg):
    """bricks.gs drops a face whose neighbour across it has a counter > 10 (brick_occupied, inc_bricks.glsl:60-62) --
    not the CPU list's >= min_voxels: with counters of exactly 10 the shared face z = 0.6 is still drawn"""
    p = peels_of(orc, pkg, [(2, 2, 3), (2, 2, 2)])
    assert abs(p[C, C, 2] - window_depth(0.6)) < 2e-6
    p = peels_of(orc, pkg, [(2, 2, 3), (2, 2, 2)], counters={(2, 2, 3): 11, (2, 2, 2): 11})
    assert abs(p[C, C, 0] - window_depth(0.8)) < 2e-6 and abs(p[C, C, 1] + window_depth(0.4)) < 2e-6
    assert abs(p[C, C, 2] - window_depth(0.4)) < 2e-6               # the only back face left
    # a full neighbour that is not in the drawn list still culls the face towards it
    p = peels_of(orc, pkg, [(2, 2, 3)], counters={(2, 2, 2): 50})
    assert abs(p[C, C, 0] - window_depth(0.8)) < 2e-6 and p[C, C, 2] == 1.0 and abs(p[C, C, 1] + window_depth(0.8)) < 2e-6


def test_depth_peels_boundary_face_is_culled_by_the_brick_the_index_wraps_to(orc, pkg):
    """bricks.gs:28-43 adds ivec3(-1, 0, 0) to the uvec3 index of brick (0, 2, 2) and linearises it (inc_bricks.glsl:25-27):
    uint arithmetic wraps to the linear id before, 59 = brick (4, 1, 2) of the 5^3 grid -- its counter decides whether
    the face on the grid's boundary is drawn.  (Seen first in the run of the shaders on Mesa, tests/test_gl_ref.py.)
    Camera on the -x side, looking along +x: the brick's x = 0 face is 2.0 from the eye, its x = 0.2 face 2.2."""
    rb = GRID[2]
    v = pkg.capi.make_view((-2.0, 0.5, 0.5), (0.5, 0.5, 0.5), (0, 1, 0), 20.0, 5, 5, (0, 0, 0), (1, 1, 1), near=NEAR, far=FAR)

    def wd(dist):
        p22, p32 = (FAR + NEAR) / (NEAR - FAR), 2 * FAR * NEAR / (NEAR - FAR)
        return (p22 * -dist + p32) / dist * 0.5 + 0.5

    def run(alias_count):
        mask = np.zeros(125, np.uint8)
        cnt = np.zeros(125, np.uint32)
        mask[(2 * rb[1] + 2) * rb[0] + 0] = 1
        cnt[(2 * rb[1] + 2) * rb[0] + 0] = 50
        assert (2 * rb[1] + 2) * rb[0] + 0 - 1 == (2 * rb[1] + 1) * rb[0] + 4 == 59
        cnt[59] = alias_count                                      # brick (4, 1, 2): not drawn, far from the ray
        return orc.depth_peels(bytes(v), GRID[0], GRID[1], rb, cnt, mask)[C, C]

    p = run(0)
    assert abs(p[0] - wd(2.0)) < 2e-6 and abs(p[1] + wd(2.2)) < 2e-6 and abs(p[2] - wd(2.2)) < 2e-6
    p = run(10)                                                    # > 10, not >= 10
    assert abs(p[0] - wd(2.0)) < 2e-6
    p = run(11)                                                    # the front face is gone: only the back face is left
    assert abs(p[0] - wd(2.2)) < 2e-6 and abs(p[1] + wd(2.2)) < 2e-6 and abs(p[2] - wd(2.2)) < 2e-6
    # an id outside the buffer (brick (2, 2, 4) towards +z: 137 >= 125) never culls: one brick seen from +z as before
    q = peels_of(orc, pkg, [(2, 2, 4)], counters={(2, 2, 3): 0})
    assert abs(q[C, C, 0] - window_depth(1.0)) < 2e-6 and abs(q[C, C, 2] - window_depth(0.8)) < 2e-6


# ---- CalibrationInverter::calculateInverseVolumes (calibration_inverter.cpp:99-155) -------------------------
def grid_lut(R=4):
    """forward LUT whose sample (sx, sy, sz) sits at ((s + 0.5) / R) of the unit box.  With the corner order of
    getCornerPoints the frustum planes (frustum.cpp:150-177) then face inwards: e.g. near = cross(e0 - e2, e3 - e2)
    = cross((-0.75, 0, 0), (-0.375, -0.375, 0)) = (0, 0, +0.28) at z = 1/8"""
    c = (np.arange(R) + 0.5) / R
    Z, Y, X = np.meshgrid(c, c, c, indexing="ij")
    return np.stack([X, Y, Z], axis=-1).astype(np.float32)


def test_inverse_volume_at_the_centre_of_eight_samples(orc):
    """voxel centres of a 2^3 volume sit in the middle of eight samples of a 4^3 LUT: equal weights, index mean
    0.5 (or 2.5) -> (idx + 0.5) / dims = 0.25 (0.75): the position itself"""
    inv = orc.inverse_volume(grid_lut(), (0, 0, 0), (1, 1, 1), (2, 2, 2))
    assert inv.shape == (2, 2, 2, 4)
    for z in range(2):
        for y in range(2):
            for x in range(2):
                exp = [0.25 + 0.5 * x, 0.25 + 0.5 * y, 0.25 + 0.5 * z, 1.0]
                assert np.allclose(inv[z, y, x], exp, atol=1e-6), (x, y, z, inv[z, y, x])


def test_inverse_volume_weights_are_inverse_distances_of_the_eight_nearest(orc):
    """a 3^3 volume: voxel 0 sits at 1/6, i.e. 1/6 of a cell from sample 0 and 5/6 from sample 1 on every axis; the
    eight nearest are {0,1}^3, weights 1 / distance (inverseDistance, :54-68)"""
    inv = orc.inverse_volume(grid_lut(), (0, 0, 0), (1, 1, 1), (3, 3, 3))
    p = np.array([1 / 6] * 3)
    num, den = np.zeros(3), 0.0
    for sx in (0, 1):
        for sy in (0, 1):
            for sz in (0, 1):
                s = (np.array([sx, sy, sz]) + 0.5) / 4
                w = 1.0 / np.linalg.norm(p - s)
                num += w * np.array([sx, sy, sz])
                den += w
    exp = (num / den + 0.5) / 4
    assert np.allclose(inv[0, 0, 0, :3], exp, atol=1e-6) and inv[0, 0, 0, 3] == 1.0
    assert np.allclose(inv[2, 2, 2, :3], 1.0 - exp, atol=1e-6)          # by symmetry
    assert abs(exp[0] - 1 / 6) > 0.01       # inverse-distance weighting of eight samples is not the trilinear inverse


def test_inverse_volume_rejects_points_outside_the_frustum(orc):
    """the hull of the LUT's corner samples is [1/8, 7/8]^3: a voxel centre outside gets (-1, -1, -1, -1) (:127-129)"""
    inv = orc.inverse_volume(grid_lut(), (0, 0, 0), (1, 1, 1), (10, 10, 10))
    assert inv[0, 0, 0].tolist() == [-1.0] * 4 and inv[9, 5, 5].tolist() == [-1.0] * 4 and inv[5, 5, 0].tolist() == [-1.0] * 4
    assert inv[5, 5, 5, 3] == 1.0 and inv[1, 1, 1, 3] == 1.0 and inv[8, 8, 8, 3] == 1.0      # centres at 0.15 / 0.85: inside
    assert inv[1, 1, 0, 3] == -1.0 and inv[9, 1, 1, 3] == -1.0                                  # 0.05 / 0.95 on one axis: outside
    # sample positions start half a voxel inside the box (:105-109): a box shifted by +1 sees nothing
    assert np.all(orc.inverse_volume(grid_lut(), (1, 1, 1), (2, 2, 2), (4, 4, 4)) == -1.0)
