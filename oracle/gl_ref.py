"""TEST INFRASTRUCTURE, build container only: RUNS the reference's GLSL -- the files where they lie under
/root/reference/glsl, compiled by Mesa's GLSL compiler and executed by llvmpipe (Mesa 23.2.1, OpenGL 4.5 compatibility
profile, no X server: oracle/gl_context.c) -- through one frame in the reference's host order and returns what the
textures hold afterwards.  tests/golden/make_gl_golden.py freezes these results as tests/golden/gl_*.npz; the oracle
(CPU suite) and the HIP path (GPU suite) are compared with them.

Unlike oracle/shader_ref.py (the shader text compiled as C++ against stand-in samplers) nothing of the GL machine is
restated here: texture filtering, texel addressing, float formats, the rasteriser's pixel centres, imageStore, SSBO
atomics, blending and the depth test are Mesa's.  What this file does restate is the reference's HOST code, i.e. which
texture is created with which format / filter and bound where, and which uniforms are set per pass:

  NetKinectArray::init / ctor         framework/NetKinectArray.cpp:42-214     formats, NEAREST filters, programs
  NetKinectArray::processDepth        framework/NetKinectArray.cpp:250-290    pre_morph.fs, mode 0 then 1 (ping-pong)
  NetKinectArray::processTextures     framework/NetKinectArray.cpp:311-428    filter / boundary / normal / quality loops
  TextureArray                        framework/rendering/TextureArray.cpp:18-36
  CalibVolumes                        framework/calibration/CalibVolumes.cpp:45-49 (BBox UBO), :76-77, :135-141 (3-D textures)
  ScreenQuad                          framework/rendering/screen_quad.cpp:8-37
  VolumeSampler                       framework/rendering/volume_sampler.cpp:9-50,86-91
  ReconIntegration                    framework/reconstruction/recon_integration.cpp:61-128 (uniforms), :243-270 (integrate),
                                      :389-404 (brick SSBO), :150-241 (draw), :280-339 (fillColors), :406-425 (depth limits),
                                      :341-354 (setVoxelSize), :474-484 (setBrickSize), :361-388 (divideBox),
                                      :431-446 (updateOccupiedBricks) -- host_grid() / occupied_bricks() below
  VolumeSampler::containedVoxels      framework/rendering/volume_sampler.cpp:50-62 (the index list of a brick)
  texture units                       source/kinect_client.cpp:245-249 (nka 1.., calibration volumes 9.., inverse 30..)

The text is compiled as it is, with two exceptions applied in memory (nothing of it is stored in this repository):
  * the shaders' helper function `sample(...)` is renamed `sample_(...)`: Mesa reserves `sample` once
    ARB_gpu_shader5 is on, and ARB_gpu_shader5 must be on for `cv_xyz[layer]` (a sampler array indexed by a uniform
    under `#version 130`, which NVIDIA's compiler takes as written);
  * `uniform mat4 gl_NormalMatrix` (tsdf_raymarch.fs:19, used by shade mode 2 only) is renamed `ref_NormalMatrix`
    and set to inverseTranspose(modelview): the built-in of that name is a mat3, so what NVIDIA binds to a mat4 of
    that name is driver-defined -- shade mode 2 of the GL run carries that assumption, the other modes do not.
Mesa's driconf switches in minigl.MESA_ENV (+ force_glsl_extensions_warn, force_compat_shaders) make its compiler
accept the remaining NVIDIA-isms without touching the text.

Never imported by the product; needs /root/reference, so it cannot run on the GPU box."""
import ctypes as C
import os
import re

import numpy as np

import minigl as m

REF_GLSL = os.environ.get("RGBDR_REFERENCE", "/root/reference") + "/glsl/"
NKA_UNIT = 1          # kinect_client.cpp:245
CV_UNIT = 9           # :247
CV_INV_UNIT = 30      # :249
IMAGE_UNIT = 3        # recon_integration.cpp:28
UNITS = {"color": NKA_UNIT, "depth": NKA_UNIT + 1, "quality": NKA_UNIT + 2, "normal": NKA_UNIT + 3, "silhouette": NKA_UNIT + 4,
         "morph_depth": NKA_UNIT + 5, "color_lab": NKA_UNIT + 6, "raw_depth": 40, "morph_input": 42}    # NetKinectArray.cpp:191-192,430-440

_gl = None
_quad = None
_programs = {}


def available():
    return os.path.exists(m.LIB_PATH) and os.path.isdir(REF_GLSL)


def _text(name):
    t = open(REF_GLSL + name).read()
    t = re.sub(r"\bsample\(", "sample_(", t)
    return t.replace("gl_NormalMatrix", "ref_NormalMatrix")


def gl():
    """the context, the named strings of Reconstruction / NetKinectArray and the screen triangle"""
    global _gl, _quad
    if _gl is None:
        os.environ.setdefault("force_glsl_extensions_warn", "true")
        os.environ.setdefault("force_compat_shaders", "true")
        g = m.GL(compat=True, version=(4, 4))
        # NetKinectArray.cpp:76,200-201; reconstruction.cpp (shading.glsl)
        for name, f in (("/bricks.glsl", "inc_bricks.glsl"), ("/inc_bbox_test.glsl", "inc_bbox_test.glsl"),
                        ("/inc_color.glsl", "inc_color.glsl"), ("/shading.glsl", "shading.glsl")):
            g.named_string(name, _text(f))
        # ScreenQuad: one triangle covering the viewport, position + texcoord interleaved
        data = np.array([-1, -1, 0, 0, 3, -1, 2, 0, -1, 3, 0, 2], np.float32)
        vao, vbo = g.gen("VertexArrays"), g.gen("Buffers")
        g.glBindVertexArray(vao)
        g.glBindBuffer(m.ARRAY_BUFFER, vbo)
        g.glBufferData(m.ARRAY_BUFFER, data.nbytes, data.ctypes.data, m.STATIC_DRAW)
        g.glEnableVertexAttribArray(0)
        g.glVertexAttribPointer(0, 2, m.FLOAT, 0, 16, C.c_void_p(0))
        g.glEnableVertexAttribArray(1)
        g.glVertexAttribPointer(1, 2, m.FLOAT, 0, 16, C.c_void_p(8))
        g.glBindVertexArray(0)
        _gl, _quad = g, vao
    return _gl


def info():
    return gl().info()


def program(name, files):
    g = gl()
    if name not in _programs:
        kinds = {"vs": m.VERTEX_SHADER, "fs": m.FRAGMENT_SHADER, "gs": m.GEOMETRY_SHADER}
        _programs[name] = g.program([g.shader(kinds[f[-2:]], _text(f), f) for f in files], name)
    return _programs[name]


class Prog:
    """uniform setters by name (globjects Program::setUniform: silently ignores inactive uniforms)"""

    def __init__(self, name, files):
        self.g, self.id, self.name = gl(), program(name, files), name

    def use(self):
        self.g.glUseProgram(self.id)

    def _loc(self, name):
        return self.g.loc(self.id, name)

    def i(self, name, v):
        l = self._loc(name)
        if l >= 0:
            self.g.glUniform1i(l, int(v))

    def u(self, name, v):
        l = self._loc(name)
        if l >= 0:
            self.g.glUniform1ui(l, int(v))

    def f(self, name, v):
        l = self._loc(name)
        if l >= 0:
            self.g.glUniform1f(l, float(np.float32(v)))

    def b(self, name, v):
        self.i(name, 1 if v else 0)

    def iv(self, name, vals):
        a = np.ascontiguousarray(vals, np.int32)
        l = self._loc(name)
        if l >= 0:
            self.g.glUniform1iv(l, a.size, a.ctypes.data)

    def fv(self, name, vals, n):
        a = np.ascontiguousarray(vals, np.float32).reshape(-1, n)
        l = self._loc(name)
        if l >= 0:
            getattr(self.g, "glUniform%dfv" % n)(l, a.shape[0], a.ctypes.data)

    def uv(self, name, vals, n):
        a = np.ascontiguousarray(vals, np.uint32).reshape(-1, n)
        l = self._loc(name)
        if l >= 0:
            getattr(self.g, "glUniform%duiv" % n)(l, a.shape[0], a.ctypes.data)

    def mat4(self, name, cols):
        """16 floats, column major (glm / gloost layout)"""
        a = np.ascontiguousarray(cols, np.float32).reshape(16)
        l = self._loc(name)
        if l >= 0:
            self.g.glUniformMatrix4fv(l, 1, 0, a.ctypes.data)


def draw_quad():
    g = gl()
    g.glBindVertexArray(_quad)
    g.glDrawArrays(m.TRIANGLES, 0, 3)
    g.glBindVertexArray(0)


def array_texture(internal, fmt, typ, W, H, layers, data=None, filt=m.LINEAR):
    g = gl()
    t = g.texture(m.TEXTURE_2D_ARRAY, filt)
    g.glPixelStorei(m.UNPACK_ALIGNMENT, 1)
    if data is not None:
        data = np.ascontiguousarray(data)
    g.glTexImage3D(m.TEXTURE_2D_ARRAY, 0, internal, W, H, layers, 0, fmt, typ, data.ctypes.data if data is not None else None)
    return t


def volume_texture(internal, fmt, vol):
    """[rz, ry, rx, ch] float32 (x fastest: calibration_volume.hpp:57-59) as a LINEAR / CLAMP_TO_EDGE 3-D texture"""
    g = gl()
    a = np.ascontiguousarray(vol, np.float32)
    t = g.texture(m.TEXTURE_3D, m.LINEAR)
    g.glPixelStorei(m.UNPACK_ALIGNMENT, 1)
    g.glTexImage3D(m.TEXTURE_3D, 0, internal, a.shape[2], a.shape[1], a.shape[0], 0, fmt, m.FLOAT, a.ctypes.data)
    return t


def bind(unit, target, tex):
    g = gl()
    g.glActiveTexture(m.TEXTURE0 + unit)
    g.glBindTexture(target, tex)


def buffer(target, data, usage=m.STATIC_DRAW):
    g = gl()
    b = g.gen("Buffers")
    a = np.ascontiguousarray(data)
    g.glBindBuffer(target, b)
    g.glBufferData(target, a.nbytes, a.ctypes.data, usage)
    return b


def check_fbo(what):
    st = gl().glCheckFramebufferStatus(m.FRAMEBUFFER)
    if st != m.FRAMEBUFFER_COMPLETE:
        raise RuntimeError("framebuffer incomplete (0x%04x) while attaching %s" % (st, what))


def draw_buffers(n):
    a = (C.c_uint * n)(*[m.COLOR_ATTACHMENT0 + i for i in range(n)])
    gl().glDrawBuffers(n, a)


def delete_textures(texs):
    a = (C.c_uint * len(texs))(*texs)
    gl().glDeleteTextures(len(texs), a)


class Calib:
    """CalibVolumes: BBox UBO at binding 2, forward volumes on units 9 + 2 i / 10 + 2 i, inverse on 30 + i"""

    def __init__(self, scene, bbox_min, bbox_max, inv_luts, inv_rgb_only=False):
        g = gl()
        self.n = scene.N
        ext = np.array([list(bbox_min) + [1.0], list(bbox_max) + [1.0]], np.float32)           # CalibVolumes.cpp:45-49
        self.ubo = buffer(m.UNIFORM_BUFFER, ext)
        g.glBindBufferBase(m.UNIFORM_BUFFER, 2, self.ubo)
        self.xyz = [volume_texture(m.RGB32F, m.RGB, scene.xyz[i]) for i in range(self.n)]       # :135-136
        self.uv = [volume_texture(m.RG32F, m.RG, scene.uv[i]) for i in range(self.n)]           # :140-141
        # :76-77 creates RGBA32F; inv_rgb_only (the 512^3 sample only, make_gl_golden.py) stores the same texels without
        # the fourth component, which no shader reads (tsdf_integration.vs:31 and tsdf_raymarch.fs take .xyz):
        # llvmpipe refuses a 512^3 RGBA32F texture (2 GiB), and takes the 1.5 GiB RGB32F one
        if inv_rgb_only:
            self.inv = [volume_texture(m.RGB32F, m.RGB, v if v.shape[-1] == 3 else np.ascontiguousarray(v[..., :3])) for v in (inv_luts or [])]
        else:
            self.inv = [volume_texture(m.RGBA32F, m.RGBA, v) for v in (inv_luts or [])]
        for i in range(self.n):                                                                  # :162-175
            bind(CV_UNIT + 2 * i, m.TEXTURE_3D, self.xyz[i])
            bind(CV_UNIT + 2 * i + 1, m.TEXTURE_3D, self.uv[i])
        for i, t in enumerate(self.inv):
            bind(CV_INV_UNIT + i, m.TEXTURE_3D, t)
        self.units_xyz = [CV_UNIT + 2 * i for i in range(self.n)]
        self.units_uv = [CV_UNIT + 2 * i + 1 for i in range(self.n)]
        self.units_inv = [CV_INV_UNIT + i for i in range(len(self.inv))]

    def free(self):
        delete_textures(self.xyz + self.uv + self.inv)


def run_frame(scene, bbox_min, bbox_max, res, inv_luts, limit=0.01, brick_size=None, res_bricks=None, limits=(0.5, 4.5),
              filter_textures=True, processed=True, refine=True, near_far=(0.5, 4.5), compress=False, keep=False,
              compress_rgb=0, use_bricks=False, min_voxels=10, inv_rgb_only=False, integrate_voxels=None):
    """One frame through the reference's shaders on Mesa; the keys of pyoracle.run_pipeline / shader_ref.run_frame.
    `scene`: rgbd_recon_amd.synth.Scene (depth [N,H,W] f32 or depth_u8, color [N,Hc,Wc,3] u8, xyz / uv forward LUTs);
    inv_luts [N][Z,Y,X,4].
    compress_rgb 1 / 5: the colour frames are scene.color_blocks [N][bytes] (DXT1 / DXT5 block streams as the server sends
    them) in GL_COMPRESSED_RGBA_S3TC_DXT{1,5}_EXT layers, decoded by the GL implementation (NetKinectArray.cpp:149-156).
    use_bricks: ReconIntegration's default mode (m_use_bricks, recon_integration.cpp:255-261): updateOccupiedBricks on
    the counters of THIS run, then one indexed draw per occupied brick with the brick's containedVoxels list.
    integrate_voxels: linear voxel ids to run tsdf_integration.vs for instead of the whole VolumeSampler (samples of grids
    too large to sweep on llvmpipe; the rest of the volume keeps the cleared -limit)."""
    g = gl()
    n = scene.N
    H, W = scene.depth.shape[1:3]
    Hc, Wc = scene.color.shape[1:3]
    cal = Calib(scene, bbox_min, bbox_max, inv_luts, inv_rgb_only=inv_rgb_only)
    vs = "texture_passthrough.vs"
    prog = {k: Prog(k, [vs, f]) for k, f in (("morph", "pre_morph.fs"), ("filter", "pre_depth.fs"), ("boundary", "pre_boundary.fs"),
                                                ("normal", "pre_normal.fs"), ("quality", "pre_quality.fs"))}

    # ---- NetKinectArray::init: textures (formats :139-172, filters :174-186) ----
    if compress_rgb:                                                                                       # m_colorArray :149-156
        blocks = np.ascontiguousarray(np.stack([np.asarray(b, np.uint8).reshape(-1) for b in scene.color_blocks]))
        fmt = m.COMPRESSED_RGBA_S3TC_DXT1_EXT if compress_rgb == 1 else m.COMPRESSED_RGBA_S3TC_DXT5_EXT
        tex_color = g.texture(m.TEXTURE_2D_ARRAY, m.LINEAR)                                                # TextureArray.cpp:27-32
        g.glPixelStorei(m.UNPACK_ALIGNMENT, 1)
        g.glCompressedTexImage3D(m.TEXTURE_2D_ARRAY, 0, fmt, Wc, Hc, n, 0, blocks.size, blocks.ctypes.data)
    else:
        tex_color = array_texture(m.RGB, m.RGB, m.UNSIGNED_BYTE, Wc, Hc, n, scene.color)                   # m_colorArray :157
    if compress:                                                                                           # :166-168
        tex_raw = array_texture(m.LUMINANCE, m.RED, m.UNSIGNED_BYTE, W, H, n, scene.depth_u8, filt=m.NEAREST)
    else:
        tex_raw = array_texture(m.LUMINANCE32F_ARB, m.RED, m.FLOAT, W, H, n, np.ascontiguousarray(scene.depth, np.float32), filt=m.NEAREST)
    tex_lab = array_texture(m.RGB32F, m.RGB, m.FLOAT, W, H, n)                                             # m_textures_color
    tex_quality = array_texture(m.LUMINANCE32F_ARB, m.RED, m.FLOAT, W, H, n)
    tex_normal = array_texture(m.RGB32F, m.RGB, m.FLOAT, W, H, n)
    tex_sil = array_texture(m.R32F, m.RED, m.FLOAT, W, H, n)
    tex_depth = array_texture(m.RG32F, m.RG, m.FLOAT, W, H, n, filt=m.NEAREST)
    tex_depth_b = array_texture(m.RG32F, m.RG, m.FLOAT, W, H, n, filt=m.NEAREST)
    depth2 = [array_texture(m.LUMINANCE32F_ARB, m.RED, m.FLOAT, W, H, n, filt=m.NEAREST) for _ in range(2)]   # front, back

    tsi = np.array([np.float32(1.0) / np.float32(W), np.float32(1.0) / np.float32(H)], np.float32)          # :195
    for k in ("filter", "normal", "quality", "morph", "boundary"):
        prog[k].use()
        prog[k].fv("texSizeInv", tsi, 2)
    prog["quality"].use()                              # CalibVolumes::getCameraPositions (host code, restated below)
    prog["quality"].fv("camera_positions", np.stack([host_camera_pos(scene.xyz[i]) for i in range(n)]), 3)
    prog["filter"].use()
    prog["filter"].i("kinect_depths", UNITS["raw_depth"])
    prog["morph"].use()
    prog["morph"].i("kinect_depths", UNITS["morph_input"])
    # setStartTextureUnit (:430-452)
    for k, names in (("filter", {"kinect_colors": "color"}), ("normal", {"kinect_depths": "depth"}),
                     ("quality", {"kinect_depths": "depth", "kinect_normals": "normal", "kinect_colors_lab": "color_lab"}),
                     ("boundary", {"kinect_colors_lab": "color_lab", "kinect_depths": "depth", "kinect_colors": "color"})):
        prog[k].use()
        for uni, unit in names.items():
            prog[k].i(uni, UNITS[unit])
    # bindToTextureUnits (:454-465)
    bind(UNITS["color"], m.TEXTURE_2D_ARRAY, tex_color)
    bind(UNITS["quality"], m.TEXTURE_2D_ARRAY, tex_quality)
    bind(UNITS["normal"], m.TEXTURE_2D_ARRAY, tex_normal)
    bind(UNITS["silhouette"], m.TEXTURE_2D_ARRAY, tex_sil)
    bind(UNITS["morph_depth"], m.TEXTURE_2D_ARRAY, depth2[0])
    bind(UNITS["color_lab"], m.TEXTURE_2D_ARRAY, tex_lab)
    bind(UNITS["raw_depth"], m.TEXTURE_2D_ARRAY, tex_raw)

    # ---- brick SSBO (recon_integration.cpp:389-401): float brick_size, pad, uvec3 resolution, pad, counters ----
    nb = int(res_bricks[0] * res_bricks[1] * res_bricks[2])
    head = np.zeros(8 + nb, np.uint32)
    head[0] = np.array([brick_size], np.float32).view(np.uint32)[0]
    head[4:7] = np.array(res_bricks, np.uint32)
    ssbo = buffer(m.SHADER_STORAGE_BUFFER, head, m.DYNAMIC_COPY)
    g.glBindBufferRange(m.SHADER_STORAGE_BUFFER, 3, ssbo, 0, head.nbytes)

    # ---- processTextures (:311-428) ----
    fbo = g.gen("Framebuffers")
    g.glDisable(m.DEPTH_TEST)
    g.glDisable(m.BLEND)
    g.glDisable(m.CULL_FACE)
    g.glViewport(0, 0, W, H)
    bind(UNITS["raw_depth"], m.TEXTURE_2D_ARRAY, tex_raw)
    g.glBindFramebuffer(m.FRAMEBUFFER, fbo)

    def attach(i, tex, layer):
        g.glFramebufferTextureLayer(m.FRAMEBUFFER, m.COLOR_ATTACHMENT0 + i, tex, 0, layer)

    # processDepth (:250-290)
    draw_buffers(1)
    bind(UNITS["morph_input"], m.TEXTURE_2D_ARRAY, tex_raw)
    pm = prog["morph"]
    pm.use()
    pm.iv("cv_xyz", cal.units_xyz)
    for mode in (0, 1):
        pm.u("mode", mode)
        if mode == 1:
            depth2.reverse()                                           # swapBuffers
            bind(UNITS["morph_input"], m.TEXTURE_2D_ARRAY, depth2[0])
        for i in range(n):
            attach(0, depth2[1], i)
            if i == 0:
                check_fbo("depth2.back (GL_LUMINANCE32F_ARB)")
            pm.u("layer", i)
            draw_quad()
    bind(UNITS["morph_input"], m.TEXTURE_2D_ARRAY, 0)
    depth2.reverse()
    bind(UNITS["morph_depth"], m.TEXTURE_2D_ARRAY, depth2[0])
    if processed:
        bind(UNITS["raw_depth"], m.TEXTURE_2D_ARRAY, depth2[0])

    # filter loop (:327-357)
    draw_buffers(2)
    pf = prog["filter"]
    pf.use()
    pf.b("filter_textures", filter_textures)
    pf.b("processed_depth", processed)
    pf.iv("cv_xyz", cal.units_xyz)
    pf.iv("cv_uv", cal.units_uv)
    near, far = np.float32(near_far[0]), np.float32(near_far[1])
    scale = np.float32(far - near)
    for i in range(n):
        pf.f("cv_min_ds", limits[0])
        pf.f("cv_max_ds", limits[1])
        attach(0, tex_depth, i)
        attach(1, tex_lab, i)
        if i == 0:
            check_fbo("depth (RG32F) + colour (RGB32F)")
        pf.u("layer", i)
        pf.b("compress", compress)
        pf.f("scale", scale)
        pf.f("near", near)
        pf.f("scaled_near", scale / np.float32(255.0))
        draw_quad()

    # boundary loop (:359-377)
    pb = prog["boundary"]
    pb.use()
    pb.iv("cv_uv", cal.units_uv)
    pb.b("refine", refine)
    bind(UNITS["depth"], m.TEXTURE_2D_ARRAY, tex_depth)
    for i in range(n):
        attach(0, tex_depth_b, i)
        attach(1, tex_sil, i)
        if i == 0:
            check_fbo("depth_b (RG32F) + silhouette (R32F)")
        pb.u("layer", i)
        draw_quad()
    bind(UNITS["depth"], m.TEXTURE_2D_ARRAY, tex_depth_b)

    # normal loop (:380-397)
    pn = prog["normal"]
    pn.use()
    pn.iv("cv_xyz", cal.units_xyz)
    pn.iv("cv_uv", cal.units_uv)
    draw_buffers(1)
    attach(1, 0, 0)
    for i in range(n):
        attach(0, tex_normal, i)
        if i == 0:
            check_fbo("normal (RGB32F)")
        pn.u("layer", i)
        draw_quad()
    g.glMemoryBarrier(m.ALL_BARRIER_BITS)

    # quality loop (:399-414)
    pq = prog["quality"]
    pq.use()
    pq.iv("cv_xyz", cal.units_xyz)
    pq.b("processed_depth", processed)
    for i in range(n):
        attach(0, tex_quality, i)
        if i == 0:
            check_fbo("quality (GL_LUMINANCE32F_ARB)")
        pq.u("layer", i)
        draw_quad()
    g.glBindFramebuffer(m.FRAMEBUFFER, 0)
    g.glUseProgram(0)
    g.glFinish()

    rd = g.read_texture
    T = m.TEXTURE_2D_ARRAY
    out = {
        "raw": list(np.ascontiguousarray(scene.depth, np.float32)),
        "morph": list(rd(T, depth2[0], m.RED, (n, H, W))),
        "depth_rg": list(rd(T, tex_depth, m.RG, (n, H, W, 2))),
        "lab": list(rd(T, tex_lab, m.RGB, (n, H, W, 3))),
        "depth_b": list(rd(T, tex_depth_b, m.RG, (n, H, W, 2))),
        "sil": list(rd(T, tex_sil, m.RED, (n, H, W))),
        "normal": list(rd(T, tex_normal, m.RGB, (n, H, W, 3))),
        "quality": list(rd(T, tex_quality, m.RED, (n, H, W))),
    }
    counters = np.zeros(8 + nb, np.uint32)
    g.glBindBuffer(m.SHADER_STORAGE_BUFFER, ssbo)
    g.glGetBufferSubData(m.SHADER_STORAGE_BUFFER, 0, counters.nbytes, counters.ctypes.data)
    out["counters"] = counters[8:].copy()

    # ---- ReconIntegration::integrate (:243-270), full sweep: VolumeSampler::sample() ----
    frame_tex = {"color": tex_color, "depth_b": tex_depth_b, "quality": tex_quality, "normal": tex_normal, "sil": tex_sil}
    vol = None
    if inv_luts is not None:
        X, Y, Z = res
        vol = g.texture(m.TEXTURE_3D, m.LINEAR)                                                    # m_volume_tsdf, :52, :347
        g.glTexImage3D(m.TEXTURE_3D, 0, m.R32F, X, Y, Z, 0, m.RED, m.FLOAT, None)
        indices = None
        if use_bricks:                      # updateOccupiedBricks (:431-446) + the occupied bricks' lists (:255-261)
            grid = host_grid(bbox_min, bbox_max, None, brick_size, res=(X, Y, Z))
            assert grid["res_bricks"] == tuple(int(v) for v in res_bricks), (grid["res_bricks"], res_bricks)
            occ = occupied_bricks(out["counters"], min_voxels)
            out["occupied"] = occ
            indices = [brick_indices(grid, int(b)) for b in occ]
        elif integrate_voxels is not None:
            indices = [np.asarray(integrate_voxels, np.uint32)]
        out["tsdf"] = integrate(cal, frame_tex, n, (X, Y, Z), limit, (W, H), volume=vol, indices=indices,
                                compact=integrate_voxels is not None)
    if keep:
        out["_gl"] = {"cal": cal, "tex": frame_tex, "volume": vol}
        delete_textures([tex_raw, tex_lab, tex_depth] + depth2)
        return out
    if vol is not None:
        delete_textures([vol])
    delete_textures([tex_color, tex_raw, tex_lab, tex_quality, tex_normal, tex_sil, tex_depth, tex_depth_b] + depth2)
    cal.free()
    return out


def bind_frame(tex):
    """what NetKinectArray::bindToTextureUnits leaves bound for ReconIntegration's programs (units 1-5)"""
    bind(UNITS["color"], m.TEXTURE_2D_ARRAY, tex["color"])
    bind(UNITS["depth"], m.TEXTURE_2D_ARRAY, tex["depth_b"])
    bind(UNITS["quality"], m.TEXTURE_2D_ARRAY, tex["quality"])
    bind(UNITS["normal"], m.TEXTURE_2D_ARRAY, tex["normal"])
    bind(UNITS["silhouette"], m.TEXTURE_2D_ARRAY, tex["sil"])


VS_PAD = 8


def integrate(cal, tex, n, res, limit, depth_wh, indices=None, volume=None, pad=None, compact=False):
    """glClearTexImage(-limit) + tsdf_integration.vs over the voxel centres of VolumeSampler (all of them, or the
    index lists of the occupied bricks) with rasteriser discard; -> [Z, Y, X] float32

    `pad`: llvmpipe's vertex-shader path (Mesa 23.2.1 draw module) returns 0 from the texture fetch that FOLLOWS a
    fetch through a dynamically indexed sampler array (`cv_xyz_inv[i]`) in every vertex whose position in the vertex
    stream is not a multiple of 8 -- the same fetch with a constant index, and the same shader in the first lane, are
    right (vs_sampler_array_bug() shows it on Mesa alone, no oracle involved).  So the voxel centres are drawn one per
    group of 8: every centre is followed by pad - 1 vertices at (2, 2, 2), whose imageStore falls outside the volume and
    is discarded (GL 4.4 section 8.26).  The shader text and the centres are unchanged; make_gl_golden.py also checks
    that pad = 8 and pad = 16 give the same volume.

    Index lists: an index past the end of the vertex buffer (a brick whose containedVoxels range leaves the z end of the
    grid) is undefined in GL without robust buffer access; the harness drops it.  Indices past the x / y end alias a
    voxel of the next row / slice through z*X*Y + y*X + x and are drawn like any other (that IS what the reference draws).
    `compact`: the vertex buffer holds only the listed voxels' centres (same floats), drawn with glDrawArrays -- for
    samples of grids whose whole VolumeSampler buffer (12 B x pad per voxel) would not fit."""
    pad = pad or VS_PAD
    g = gl()
    X, Y, Z = res
    p = Prog("integration", ["tsdf_integration.vs"])
    assert g.glGetAttribLocation(p.id, b"in_Position") == 0
    p.use()
    p.iv("cv_xyz_inv", cal.units_inv)                                   # :92
    p.i("volume_tsdf", IMAGE_UNIT)                                      # :95
    p.i("kinect_colors", 1)
    p.i("kinect_depths", 2)
    p.i("kinect_qualities", 3)
    p.i("kinect_normals", 4)
    p.i("kinect_silhouettes", 5)
    p.u("num_kinects", n)
    p.uv("res_depth", np.array(depth_wh, np.uint32), 2)
    p.f("limit", limit)
    p.uv("res_tsdf", np.array([X, Y, Z], np.uint32), 3)                 # setVoxelSize :346
    bind_frame(tex)
    # VolumeSampler: (x + .5) * step, x fastest (volume_sampler.cpp:14-24)
    sx, sy, sz = np.float32(1.0) / np.float32(X), np.float32(1.0) / np.float32(Y), np.float32(1.0) / np.float32(Z)
    if compact:
        ids = np.concatenate([np.asarray(i, np.int64) for i in indices])
        ids = ids[ids < X * Y * Z]
        zz, rem = np.divmod(ids, X * Y)
        yy, xx = np.divmod(rem, X)
        pos = np.stack([(xx.astype(np.float32) + np.float32(0.5)) * sx, (yy.astype(np.float32) + np.float32(0.5)) * sy,
                        (zz.astype(np.float32) + np.float32(0.5)) * sz], axis=-1).astype(np.float32)
        nvert = ids.size
    else:
        pos = np.empty((Z, Y, X, 3), np.float32)
        pos[..., 0] = ((np.arange(X, dtype=np.float32) + np.float32(0.5)) * sx)[None, None, :]
        pos[..., 1] = ((np.arange(Y, dtype=np.float32) + np.float32(0.5)) * sy)[None, :, None]
        pos[..., 2] = ((np.arange(Z, dtype=np.float32) + np.float32(0.5)) * sz)[:, None, None]
        nvert = X * Y * Z
    if pad > 1:
        padded = np.full((nvert, pad, 3), 2.0, np.float32)
        padded[:, 0] = pos.reshape(-1, 3)
        pos = padded
    vao = g.gen("VertexArrays")
    g.glBindVertexArray(vao)
    vbo = buffer(m.ARRAY_BUFFER, pos)
    g.glEnableVertexAttribArray(0)
    g.glVertexAttribPointer(0, 3, m.FLOAT, 0, 12, C.c_void_p(0))
    vol = volume
    if vol is None:
        vol = g.texture(m.TEXTURE_3D, m.LINEAR)
        g.glTexImage3D(m.TEXTURE_3D, 0, m.R32F, X, Y, Z, 0, m.RED, m.FLOAT, None)                 # :347
    g.glEnable(m.RASTERIZER_DISCARD)
    neg = C.c_float(-float(np.float32(limit)))
    g.glClearTexImage(vol, 0, m.RED, m.FLOAT, C.byref(neg))
    g.glBindImageTexture(IMAGE_UNIT, vol, 0, 1, 0, m.WRITE_ONLY, m.R32F)
    if indices is None or compact:
        g.glDrawArrays(m.POINTS, 0, nvert * pad)
    else:
        for idx in indices:                                             # m_sampler.sample(m_bricks[index].indices), :258-260
            idx = np.asarray(idx, np.uint32)
            idx = idx[idx < np.uint32(X * Y * Z)]
            if idx.size == 0:
                continue
            a = (idx[:, None] * np.uint32(pad) + np.arange(pad, dtype=np.uint32)[None, :]).reshape(-1)
            a = np.ascontiguousarray(a, np.uint32)
            g.glDrawElements(m.POINTS, a.size, m.UNSIGNED_INT, a.ctypes.data)
    g.glMemoryBarrier(m.ALL_BARRIER_BITS)
    g.glDisable(m.RASTERIZER_DISCARD)
    g.glUseProgram(0)
    g.glBindVertexArray(0)
    g.glFinish()
    tsdf = g.read_texture(m.TEXTURE_3D, vol, m.RED, (Z, Y, X))
    if volume is None:
        delete_textures([vol])
    b = (C.c_uint * 1)(vbo)
    g.glBindBuffer(m.ARRAY_BUFFER, 0)
    C.CFUNCTYPE(None, C.c_int, C.c_void_p)(g._lib.glctx_proc(b"glDeleteBuffers"))(1, b)
    return tsdf


# ---- ReconIntegration's grid and bricks: HOST code of the reference, restated in binary32 like glm::fvec3 does it ----
_F = np.float32


def host_grid(bbox_min, bbox_max, voxel_size, brick_size, res=None):
    """setVoxelSize (recon_integration.cpp:341-354): m_res_volume = ceil(extent / voxel);  setBrickSize (:474-484):
    m_brick_size = voxel * round(size / voxel);  divideBox (:361-388): bricks x-fastest from bbox.min in steps of
    m_brick_size accumulated in float, the last one per axis clipped, each with the index list of
    VolumeSampler::containedVoxels (volume_sampler.cpp:50-62).  Membership is separable, so per axis the bricks'
    (first voxel, bound) pairs are kept and a brick's list is built from them in the reference's loop order.
    voxel_size None: `brick_size` is already the adjusted m_brick_size and `res` the volume resolution."""
    mn = [_F(v) for v in bbox_min]
    ext = [_F(_F(bbox_max[a]) - mn[a]) for a in range(3)]                  # getPMax()[a] - getPMin()[a], mathType = float
    if voxel_size is not None:
        vs = _F(voxel_size)
        res = tuple(int(np.ceil(_F(ext[a] / vs))) for a in range(3))
        q = _F(_F(brick_size) / vs)
        bs = _F(vs * _F(np.floor(np.abs(q) + _F(0.5)) * np.sign(q)))      # glm::round: half away from zero
    else:
        bs = _F(brick_size)
    axes = []
    for a in range(3):
        size, lo0 = ext[a], mn[a]
        step = _F(_F(1.0) / _F(res[a]))                                    # glm::fvec3 step{1.0f / fvec3{m_dimensions}}
        start, rng = lo0, []
        while _F(_F(size - start) + lo0) > _F(0.0):                        # while(size.x - start.x + min.x > 0.0f)
            bsz = min(bs, _F(_F(size - start) + lo0))                      # glm::min(fvec3{m_brick_size}, size - start + min)
            pos_n = _F(_F(start - lo0) / size)                             # (curr_brick.pos - min) / size
            size_n = _F(bsz / size)                                        # curr_brick.size / size
            first = int(_F(pos_n / step))                                  # unsigned x = pos.x / step.x
            bound = _F(_F(pos_n + size_n) / step)                          # x < (pos.x + size.x) / step.x
            last = first
            while _F(last) < bound:
                last += 1
            rng.append((first, last))                                      # voxels first .. last - 1
            start = _F(start + bs)                                         # start.x += m_brick_size
        axes.append(rng)
    return {"res": tuple(res), "brick_size": float(bs), "res_bricks": tuple(len(r) for r in axes), "axes": axes}


def brick_indices(grid, brick):
    """m_bricks[brick].indices: for y, for x, for z: z * X * Y + y * X + x  (volume_sampler.cpp:53-58; unsigned arithmetic)"""
    X, Y, Z = grid["res"]
    rx, ry, rz = grid["res_bricks"]
    bx, by, bz = brick % rx, (brick // rx) % ry, brick // (rx * ry)
    xs = np.arange(*grid["axes"][0][bx], dtype=np.uint64)
    ys = np.arange(*grid["axes"][1][by], dtype=np.uint64)
    zs = np.arange(*grid["axes"][2][bz], dtype=np.uint64)
    ids = zs[None, None, :] * np.uint64(X * Y) + ys[:, None, None] * np.uint64(X) + xs[None, :, None]
    return (ids.reshape(-1) & np.uint64(0xFFFFFFFF)).astype(np.uint32)


def host_camera_pos(cv_xyz):
    """CalibVolumes::getCameraPositions = Frustum(getCornerPoints(cv_xyz)).getCameraPos() (CalibVolumes.cpp:98-113,224-230;
    frustum.cpp:21-33, closestPoint :97-111), in binary32 with glm's association (dot = x*x + y*y + z*z, left to right).
    The harness's own restatement of this host code: the run on Mesa does not borrow it from the oracle."""
    v = np.asarray(cv_xyz, np.float32)
    ez, ey, ex = v.shape[0] - 1, v.shape[1] - 1, v.shape[2] - 1
    c = [v[0, 0, 0], v[0, ey, 0], v[0, ey, ex], v[0, 0, ex], v[ez, 0, 0], v[ez, ey, 0], v[ez, ey, ex], v[ez, 0, ex]]   # curr_volume(x, y, z)
    c = [np.asarray(p[:3], np.float32) for p in c]
    four = _F(4.0)

    def dot(a, b):
        return _F(_F(_F(a[0] * b[0]) + _F(a[1] * b[1])) + _F(a[2] * b[2]))

    def closest(p, u, q, w):
        w0 = p - q
        a, b, cc, d, e = dot(u, u), dot(u, w), dot(w, w), dot(u, w0), dot(w, w0)
        den = _F(_F(a * cc) - _F(b * b))
        sc = _F(_F(_F(b * e) - _F(cc * d)) / den)
        tc = _F(_F(_F(a * e) - _F(b * d)) / den)
        return ((p + u * sc) + (q + w * tc)) * _F(0.5)
    near = (((c[0] + c[1]) + c[2]) + c[3]) / four
    far = (((c[4] + c[5]) + c[6]) + c[7]) / four
    view_dir = far - near
    pts = [closest(c[i], c[i] - c[i + 4], near, view_dir) for i in range(4)]
    return ((((pts[0] + pts[1]) + pts[2]) + pts[3]) / four).astype(np.float32)


def occupied_bricks(counters, min_voxels):
    """updateOccupiedBricks (:436-440): ids with m_active_bricks[i] >= m_min_voxels_per_brick, ascending"""
    return np.nonzero(np.asarray(counters, np.uint32) >= np.uint32(min_voxels))[0].astype(np.uint32)


def release(out):
    """frees what run_frame(keep=True) left alive"""
    st = out.pop("_gl", None)
    if st:
        delete_textures(list(st["tex"].values()) + ([st["volume"]] if st["volume"] else []))
        st["cal"].free()


_TF_BUFFER, _INTERLEAVED = 0x8C8E, 0x8C8C


def vs_sampler_array_bug(count=64):
    """Mesa against itself, no reference text and no oracle involved: a vertex shader of this file's own fetches a
    coordinate from a sampler3D array -- once with a uniform index, once with the constant 0 (the uniform IS 0) -- and
    uses it for a second fetch from an all-ones LINEAR array texture.  On llvmpipe 23.2.1 the uniform-index variant
    returns 0 for every vertex that is not the first of a group of 8.  -> list of the vertex numbers that are wrong
    ([] on a correct implementation); integrate() pads its vertex stream accordingly."""
    g = gl()
    vol = volume_texture(m.RGBA32F, m.RGBA, np.full((4, 4, 4, 4), 0.5, np.float32))
    ones = array_texture(m.R32F, m.RED, m.FLOAT, 8, 8, 2, np.ones((2, 8, 8), np.float32))
    res = []
    for index in ("k", "0"):
        src = ("#version 430\nin vec3 p; uniform sampler3D[5] vols; uniform sampler2DArray img; uniform uint k; out float v;\n"
               "void main() { vec3 c = texture(vols[%s], p).xyz; v = texture(img, vec3(c.xy, 0.0)).r; gl_Position = vec4(0.0, 0.0, 0.0, 1.0); }" % index)
        prog = g.glCreateProgram()
        g.glAttachShader(prog, g.shader(m.VERTEX_SHADER, src, "probe"))
        names = (C.c_char_p * 1)(b"v")
        C.CFUNCTYPE(None, C.c_uint, C.c_int, C.c_void_p, C.c_uint)(g._lib.glctx_proc(b"glTransformFeedbackVaryings"))(prog, 1, names, _INTERLEAVED)
        g.glLinkProgram(prog)
        g.glUseProgram(prog)
        bind(CV_INV_UNIT, m.TEXTURE_3D, vol)
        bind(UNITS["silhouette"], m.TEXTURE_2D_ARRAY, ones)
        units = np.full(5, CV_INV_UNIT, np.int32)
        g.glUniform1iv(g.loc(prog, "vols"), 5, units.ctypes.data)
        g.glUniform1i(g.loc(prog, "img"), UNITS["silhouette"])
        if g.loc(prog, "k") >= 0:
            g.glUniform1ui(g.loc(prog, "k"), 0)
        pts = np.random.default_rng(5).random((count, 3), dtype=np.float32)
        vao = g.gen("VertexArrays")
        g.glBindVertexArray(vao)
        buffer(m.ARRAY_BUFFER, pts)
        g.glEnableVertexAttribArray(0)
        g.glVertexAttribPointer(0, 3, m.FLOAT, 0, 12, C.c_void_p(0))
        out = np.zeros(count, np.float32)
        tfb = buffer(_TF_BUFFER, out, m.DYNAMIC_COPY)
        g.glBindBufferBase(_TF_BUFFER, 0, tfb)
        g.glEnable(m.RASTERIZER_DISCARD)
        C.CFUNCTYPE(None, C.c_uint)(g._lib.glctx_proc(b"glBeginTransformFeedback"))(m.POINTS)
        g.glDrawArrays(m.POINTS, 0, count)
        C.CFUNCTYPE(None)(g._lib.glctx_proc(b"glEndTransformFeedback"))()
        g.glDisable(m.RASTERIZER_DISCARD)
        g.glFinish()
        g.glBindBuffer(_TF_BUFFER, tfb)
        g.glGetBufferSubData(_TF_BUFFER, 0, out.nbytes, out.ctypes.data)
        g.glBindBufferBase(_TF_BUFFER, 0, 0)
        g.glUseProgram(0)
        g.glBindVertexArray(0)
        res.append(out)
    delete_textures([vol, ones])
    assert np.all(res[1] == 1.0), "the constant-index variant is wrong too"
    return [int(i) for i in np.nonzero(res[0] != res[1])[0]]


# ---- consumers of the volume: ReconIntegration::drawF = drawDepthLimits + draw + fillColors --------------
CUBE = np.array([1, 1, 1, 0, 1, 1, 1, 1, 0, 0, 1, 0, 1, 0, 1, 0, 0, 1, 0, 0, 0, 1, 0, 0], np.float32)     # unit_cube.cpp:20-29
CUBE_STRIP = np.array([3, 2, 6, 7, 4, 2, 0, 3, 1, 6, 5, 4, 1, 0], np.uint8)                               # :45-48
_cube_vao = None


def cube_vao():
    global _cube_vao
    if _cube_vao is None:
        g = gl()
        _cube_vao = g.gen("VertexArrays")
        g.glBindVertexArray(_cube_vao)
        buffer(m.ARRAY_BUFFER, CUBE)
        g.glEnableVertexAttribArray(0)
        g.glVertexAttribPointer(0, 3, m.FLOAT, 0, 12, C.c_void_p(0))
        g.glBindVertexArray(0)
    return _cube_vao


def lod_layout(W, H):
    """ViewLod::setResolution (view_lod.cpp:24-48): -> (full width, offsets [n,2], resolutions [n,2])"""
    n = 1 + int(np.floor(np.log2(np.float32(min(W, H)))))
    FW = int(np.float32(W) * np.float32(1.5))
    off, res = np.zeros((n, 2), np.uint32), np.zeros((n, 2), np.uint32)
    ox, oy = W, H
    for i in range(n):
        p = np.power(np.float32(2.0), np.float32(i))
        res[i] = (int(np.floor(np.float32(W) / p)), int(np.floor(np.float32(H) / p)))
        if i > 0:
            oy -= int(res[i][1])
            off[i] = (ox, oy)
    return FW, off, res


class ViewLod:
    """view_lod.cpp: one RGBA32F colour atlas (LINEAR, MIRRORED_REPEAT s/t) + DEPTH_COMPONENT32 (NEAREST), 1.5 W x H"""

    def __init__(self, W, H):
        g = gl()
        self.W, self.H = W, H
        self.FW, self.off, self.res = lod_layout(W, H)
        half = np.full((H, self.FW, 4), 0.5, np.float32)
        self.color = g.texture(m.TEXTURE_2D, m.LINEAR)
        g.glTexImage2D(m.TEXTURE_2D, 0, m.RGBA32F, self.FW, H, 0, m.RGBA, m.FLOAT, half.ctypes.data)
        g.glTexParameteri(m.TEXTURE_2D, m.TEXTURE_WRAP_S, m.MIRRORED_REPEAT)
        g.glTexParameteri(m.TEXTURE_2D, m.TEXTURE_WRAP_T, m.MIRRORED_REPEAT)
        self.depth = g.texture(m.TEXTURE_2D, m.NEAREST)
        g.glTexImage2D(m.TEXTURE_2D, 0, m.DEPTH_COMPONENT32, self.FW, H, 0, m.DEPTH_COMPONENT, m.FLOAT, half[..., 0].copy().ctypes.data)
        self.fbo = g.gen("Framebuffers")
        g.glBindFramebuffer(m.FRAMEBUFFER, self.fbo)
        g.glFramebufferTexture2D(m.FRAMEBUFFER, m.COLOR_ATTACHMENT0, m.TEXTURE_2D, self.color, 0)
        g.glFramebufferTexture2D(m.FRAMEBUFFER, m.DEPTH_ATTACHMENT, m.TEXTURE_2D, self.depth, 0)
        draw_buffers(1)
        check_fbo("ViewLod (RGBA32F + DEPTH_COMPONENT32)")
        g.glBindFramebuffer(m.FRAMEBUFFER, 0)

    def enable(self, lod=0, clear_color=True, clear_depth=True):
        g = gl()
        g.glBindFramebuffer(m.FRAMEBUFFER, self.fbo)
        g.glViewport(int(self.off[lod][0]), int(self.off[lod][1]), int(self.res[lod][0]), int(self.res[lod][1]))
        if clear_color:
            g.glClearColor(0.0, 1.0, 0.0, 0.0)
            g.glClear(m.COLOR_BUFFER_BIT)
        if clear_depth:
            g.glClear(m.DEPTH_BUFFER_BIT)

    def disable(self):
        gl().glBindFramebuffer(m.FRAMEBUFFER, 0)

    def bind_units(self, start):
        bind(start, m.TEXTURE_2D, self.color)
        bind(start + 1, m.TEXTURE_2D, self.depth)

    def read(self, full=False):
        g = gl()
        c = g.read_texture(m.TEXTURE_2D, self.color, m.RGBA, (self.H, self.FW, 4))
        d = g.read_texture(m.TEXTURE_2D, self.depth, m.DEPTH_COMPONENT, (self.H, self.FW))
        return (c, d) if full else (c[:, :self.W].copy(), d[:, :self.W].copy())

    def free(self):
        delete_textures([self.color, self.depth])


def set_matrices(view):
    """g_camera.set() / update_model_matrix(): the fixed-function matrix stacks the shaders read as gl_ModelViewMatrix /
    gl_ProjectionMatrix (compatibility profile)"""
    g = gl()
    pr = np.array(list(view.projection), np.float32)
    mv = np.array(list(view.modelview), np.float32)
    g.glMatrixMode(m.PROJECTION)
    g.glLoadMatrixf(pr.ctypes.data)
    g.glMatrixMode(m.MODELVIEW)
    g.glLoadMatrixf(mv.ctypes.data)
    return mv, pr


def brick_buffers(brick_size, res_bricks, counters, min_voxels):
    """SSBO 3 as divideBox lays it out with the frame's counters in it, SSBO 4 = updateOccupiedBricks' id list
    (recon_integration.cpp:389-401, :428-446) -> (ssbo3, ssbo4, occupied ids)"""
    g = gl()
    nb = int(np.prod(res_bricks))
    head = np.zeros(8 + nb, np.uint32)
    head[0] = np.array([brick_size], np.float32).view(np.uint32)[0]
    head[4:7] = np.array(res_bricks, np.uint32)
    head[8:] = counters
    b3 = buffer(m.SHADER_STORAGE_BUFFER, head, m.DYNAMIC_COPY)
    g.glBindBufferRange(m.SHADER_STORAGE_BUFFER, 3, b3, 0, head.nbytes)
    ids = np.nonzero(np.asarray(counters) >= min_voxels)[0].astype(np.uint32)
    occ = np.zeros(8 + nb, np.uint32)
    occ[:ids.size] = ids
    b4 = buffer(m.SHADER_STORAGE_BUFFER, occ, m.DYNAMIC_DRAW)
    if ids.size:
        g.glBindBufferRange(m.SHADER_STORAGE_BUFFER, 4, b4, 0, 4 * ids.size)
    return b3, b4, ids


def depth_limits(view, brick_size, res_bricks, counters, min_voxels):
    """ReconIntegration::drawDepthLimits (:406-425): the occupied bricks as instanced unit cubes through bricks.{vs,gs,fs}
    into an RGBA32F target without depth buffer, cleared to (1, 0, 1, 0), MIN blending -> [H, W, 4]"""
    g = gl()
    W, H = view.width, view.height
    tex = g.texture(m.TEXTURE_2D, m.LINEAR)
    g.glTexImage2D(m.TEXTURE_2D, 0, m.RGBA32F, W, H, 0, m.RGBA, m.UNSIGNED_BYTE, None)                 # view.cpp:41
    fbo = g.gen("Framebuffers")
    g.glBindFramebuffer(m.FRAMEBUFFER, fbo)
    g.glFramebufferTexture2D(m.FRAMEBUFFER, m.COLOR_ATTACHMENT0, m.TEXTURE_2D, tex, 0)
    draw_buffers(1)
    check_fbo("depth-limit view (RGBA32F)")
    g.glViewport(0, 0, W, H)
    g.glClearColor(1.0, 0.0, 1.0, 0.0)                                                                 # :144
    g.glClear(m.COLOR_BUFFER_BIT)
    b3, b4, ids = brick_buffers(brick_size, res_bricks, counters, min_voxels)
    p = Prog("bricks", ["bricks.vs", "bricks.fs", "bricks.gs"])
    assert g.glGetAttribLocation(p.id, b"in_Position") == 0
    p.use()
    set_matrices(view)
    g.glEnable(m.DEPTH_TEST)
    g.glDepthFunc(m.LESS)
    g.glDisable(m.CULL_FACE)
    g.glEnable(m.BLEND)
    g.glBlendEquation(m.MIN)
    if ids.size:
        g.glBindVertexArray(cube_vao())
        g.glDrawElementsInstanced(m.TRIANGLE_STRIP, CUBE_STRIP.size, m.UNSIGNED_BYTE, CUBE_STRIP.ctypes.data, int(ids.size))
        g.glBindVertexArray(0)
    g.glUseProgram(0)
    g.glDisable(m.BLEND)
    g.glBlendEquation(m.FUNC_ADD)
    g.glBindFramebuffer(m.FRAMEBUFFER, 0)
    g.glFinish()
    out = g.read_texture(m.TEXTURE_2D, tex, m.RGBA, (H, W, 4))
    return out, tex


def raymarch(view, cal, tex, volume, n, limit, peels_tex=None, target=None):
    """ReconIntegration::draw (:170-241) into a ViewLod (what m_fill_holes = true renders to; the same shader output
    the default framebuffer would quantise to RGBA8): tsdf_raymarch.{vs,fs} over UnitCube::draw
    -> (colour [H,W,4], depth [H,W], sample counts [H,W], the ViewLod)"""
    g = gl()
    W, H = view.width, view.height
    p = Prog("raymarch", ["tsdf_raymarch.vs", "tsdf_raymarch.fs"])
    p.use()
    mv, pr = set_matrices(view)
    # ctor (:61-87)
    p.mat4("vol_to_world", list(view.vol_to_world))
    p.i("kinect_colors", 1)
    p.i("kinect_depths", 2)
    p.i("kinect_qualities", 3)
    p.i("kinect_normals", 4)
    p.iv("cv_xyz_inv", cal.units_inv)
    p.iv("cv_uv", cal.units_uv)
    p.u("num_kinects", n)
    p.f("limit", limit)
    p.i("depth_peels", 17)
    p.b("skipSpace", bool(view.skip_space))
    p.i("tex_num_samples", IMAGE_UNIT + 1)
    p.fv("viewport_offset", [0.0, 0.0], 2)
    p.i("volume_tsdf", 29)
    # draw (:177-209)
    p.mat4("img_to_eye_curr", list(view.img_to_eye))
    p.mat4("NormalMatrix", list(view.normal_matrix))
    p.fv("CameraPos", list(view.camera_pos), 3)
    # the mat4 the text declares as gl_NormalMatrix (see the header of this file): inverseTranspose(modelview)
    nm = np.linalg.inv(mv.reshape(4, 4).T.astype(np.float64)).T                   # row-major maths on M = mv^T-of-columns
    p.mat4("ref_NormalMatrix", np.ascontiguousarray(nm.T.reshape(-1), np.float32))
    settings = buffer(m.UNIFORM_BUFFER, np.array([view.shade_mode, 0, 0, 0], np.int32))     # kinect_client.cpp:262-266
    g.glBindBufferBase(m.UNIFORM_BUFFER, 1, settings)
    bind_frame(tex)
    bind(29, m.TEXTURE_3D, volume)                                                           # :348
    if peels_tex is not None:
        bind(17, m.TEXTURE_2D, peels_tex)
    ns = g.texture(m.TEXTURE_2D, m.LINEAR)
    g.glTexImage2D(m.TEXTURE_2D, 0, m.R32F, W, H, 0, m.RED, m.FLOAT, None)                   # resize :498
    zero = C.c_float(0.0)
    g.glClearTexImage(ns, 0, m.RED, m.FLOAT, C.byref(zero))
    g.glBindImageTexture(IMAGE_UNIT + 1, ns, 0, 0, 0, m.WRITE_ONLY, m.R32F)
    t = target or ViewLod(W, H)
    g.glEnable(m.DEPTH_TEST)
    g.glDepthFunc(m.LESS)
    t.enable()
    g.glDisable(m.CULL_FACE)
    g.glBindVertexArray(cube_vao())
    g.glDrawElements(m.TRIANGLE_STRIP, CUBE_STRIP.size, m.UNSIGNED_BYTE, CUBE_STRIP.ctypes.data)
    g.glBindVertexArray(0)
    g.glMemoryBarrier(m.ALL_BARRIER_BITS)
    g.glUseProgram(0)
    t.disable()
    g.glFinish()
    color, depth = t.read()
    samples = g.read_texture(m.TEXTURE_2D, ns, m.RED, (H, W))
    delete_textures([ns])
    return color, depth, samples, t


def fill_colors(a, W, H):
    """ReconIntegration::fillColors (:280-339) on the ViewLod `a` the ray-march rendered into; the last pass goes to an
    RGBA32F + DEPTH_COMPONENT32 target of this file instead of the window.  -> (filled colour [H,W,4], depth [H,W])"""
    g = gl()
    b = ViewLod(W, H)
    nl = a.off.shape[0]
    vs = "texture_passthrough.vs"
    transfer, inpaint, colorfill = Prog("transfer", [vs, "framebuffer_transfer.fs"]), Prog("inpaint", [vs, "tsdf_inpaint.fs"]), Prog("colorfill", [vs, "tsdf_colorfill.fs"])
    rinv = np.array([np.float32(1.0) / np.float32(a.FW), np.float32(1.0) / np.float32(H)], np.float32)
    # uvec2[20] arrays of which the reference uploads numLods elements (:502-510).  tsdf_colorfill.fs reads levels + 1 and + 2
    # past num_lods: zeros in a freshly linked program -- what the oracle and the library evaluate --, but whatever an earlier,
    # larger window left there after a resize.  The programs of this harness are cached across calls, so all 20 elements
    # are written: every call sees the fresh program of the reference's start-up.
    off20, res20 = np.zeros((20, 2), np.uint32), np.zeros((20, 2), np.uint32)
    off20[:nl], res20[:nl] = a.off, a.res
    for p in (inpaint, colorfill):                                                             # ctor :103-128, resize :502-510
        p.use()
        p.i("texture_color", 15)
        p.i("texture_depth", 16)
        p.fv("viewport_offset", [0.0, 0.0], 2)
        p.uv("texture_offsets", off20, 2)
        p.uv("texture_resolutions", res20, 2)
        p.fv("resolution_inv", rinv, 2)
    colorfill.i("num_lods", nl)
    transfer.use()
    transfer.i("texture_color", 15)
    transfer.i("texture_depth", 16)
    transfer.uv("resolution_tex", np.array([a.FW, H], np.uint32), 2)
    g.glEnable(m.DEPTH_TEST)
    g.glDepthFunc(m.ALWAYS)

    def do_transfer():
        nonlocal a, b
        a.bind_units(15)
        b.enable(0)
        transfer.use()
        transfer.i("lod", 0)
        draw_quad()
        b.disable()
        a, b = b, a

    do_transfer()
    for i in range(1, nl):
        a.bind_units(15)
        b.enable(i, False, False)
        inpaint.use()
        inpaint.i("lod", i - 1)
        draw_quad()
        b.disable()
        a, b = b, a
        do_transfer()
    g.glDepthFunc(m.LESS)
    b.bind_units(15)                                       # m_view_inpaint2 (:316)
    win = ViewLod(W, H)                                    # stands for the window: only its LOD-0 rectangle is used
    win.enable(0)
    g.glClearColor(0.0, 0.0, 0.0, 0.0)
    g.glClear(m.COLOR_BUFFER_BIT | m.DEPTH_BUFFER_BIT)     # kinect_client.cpp:614
    colorfill.use()
    colorfill.uv("resolution_tex", a.res[0], 2)
    colorfill.i("lod", 0)
    draw_quad()
    win.disable()
    g.glUseProgram(0)
    g.glFinish()
    oc, od = win.read()
    atlas = b.read(full=True)[0]
    win.free()
    a.free()
    b.free()
    return oc, od, atlas


def run_views(name, scene, cfg, geo, inv, out):
    """every view of shader_cases.VIEW_CASES[name] from the frame run_frame(keep=True) left in GL"""
    import pyoracle
    import shader_cases
    from rgbd_recon_amd import capi, synth
    st = out["_gl"]
    n = scene.N
    res = {}
    for key, eye, mode, skip, fill in shader_cases.VIEW_CASES[name]:
        view = shader_cases.make_view(capi, synth, eye, mode, skip)
        v = pyoracle.View.from_buffer_copy(bytes(view))
        peels_tex = None
        if skip:
            peels, peels_tex = depth_limits(v, geo.brick_size, tuple(geo.res_bricks), out["counters"], cfg.min_voxels_per_brick)
            res[key + "_peels"] = peels
        color, depth, ns, target = raymarch(v, st["cal"], st["tex"], st["volume"], n, cfg.tsdf_limit, peels_tex)
        res[key + "_color"], res[key + "_depth"], res[key + "_samples"] = color, depth, ns
        if fill:
            fc, fd, atlas = fill_colors(target, v.width, v.height)
            res[key + "_filled_color"], res[key + "_filled_depth"] = fc, fd
        else:
            target.free()
        if peels_tex is not None:
            delete_textures([peels_tex])
    return res


def decode_dxt(blocks, W, H, mode):
    """What the GL driver makes of a DXT1 / DXT5 colour frame -- the reference uploads the server's blocks as
    GL_COMPRESSED_RGBA_S3TC_DXT{1,5}_EXT layers of m_colorArray (NetKinectArray.cpp:149-156, TextureArray.cpp:30-32)
    and lets the texture unit decode them.  blocks: the block stream of one layer -> [H, W, 4] u8 as Mesa decodes it."""
    g = gl()
    b = np.ascontiguousarray(blocks, np.uint8).reshape(-1)
    fmt = m.COMPRESSED_RGBA_S3TC_DXT1_EXT if mode == 1 else m.COMPRESSED_RGBA_S3TC_DXT5_EXT
    t = g.texture(m.TEXTURE_2D_ARRAY, m.LINEAR)
    g.glPixelStorei(m.UNPACK_ALIGNMENT, 1)
    g.glCompressedTexImage3D(m.TEXTURE_2D_ARRAY, 0, fmt, W, H, 1, 0, b.size, b.ctypes.data)
    out = g.read_texture(m.TEXTURE_2D_ARRAY, t, m.RGBA, (1, H, W, 4), dtype=np.uint8, gltype=m.UNSIGNED_BYTE)
    delete_textures([t])
    return out[0]
