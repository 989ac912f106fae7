"""TEST INFRASTRUCTURE, build container only: a small ctypes binding of the OpenGL entry points oracle/gl_ref.py needs,
on the windowless Mesa llvmpipe context of oracle/gl_context.c (oracle/_ref/libglctx.so, `make -C oracle glctx`).
Nothing here is used by the product or on the GPU box."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_ref", "libglctx.so")

# Mesa's switches for GLSL that only NVIDIA's compiler takes as written (the reference was developed on NVIDIA):
# `uniform mat4 gl_ModelViewMatrix;` / `out float gl_FragDepth;` redeclarations (tsdf_raymarch.fs:17-19,40),
# #extension lines after the first declaration (inc_bricks.glsl:4-6 is #included below declarations).  They change
# what the compiler ACCEPTS, not what the text computes; the shader text itself is compiled unmodified.
MESA_ENV = {
    "allow_glsl_builtin_variable_redeclaration": "true",
    "allow_glsl_extension_directive_midshader": "true",
    "LP_NUM_THREADS": "8",
}

# enums (GL/glcorearb.h)
TEXTURE_2D, TEXTURE_3D, TEXTURE_2D_ARRAY = 0x0DE1, 0x806F, 0x8C1A
TEXTURE_MIN_FILTER, TEXTURE_MAG_FILTER = 0x2801, 0x2800
TEXTURE_WRAP_S, TEXTURE_WRAP_T, TEXTURE_WRAP_R = 0x2802, 0x2803, 0x8072
NEAREST, LINEAR = 0x2600, 0x2601
CLAMP_TO_EDGE, MIRRORED_REPEAT, REPEAT = 0x812F, 0x8370, 0x2901
RED, RG, RGB, RGBA = 0x1903, 0x8227, 0x1907, 0x1908
RED_INTEGER = 0x8D94
R32F, RG32F, RGB32F, RGBA32F = 0x822E, 0x8230, 0x8815, 0x8814
R32UI = 0x8236
RGB8, RGBA8 = 0x8051, 0x8058
LUMINANCE32F_ARB, LUMINANCE = 0x8818, 0x1909
DEPTH_COMPONENT, DEPTH_COMPONENT32, DEPTH_COMPONENT24, DEPTH_COMPONENT32F = 0x1902, 0x81A7, 0x81A6, 0x8CAC
COMPRESSED_RGBA_S3TC_DXT1_EXT, COMPRESSED_RGBA_S3TC_DXT5_EXT = 0x83F1, 0x83F3
FLOAT, UNSIGNED_BYTE, UNSIGNED_INT = 0x1406, 0x1401, 0x1405
FRAMEBUFFER, COLOR_ATTACHMENT0, DEPTH_ATTACHMENT = 0x8D40, 0x8CE0, 0x8D00
FRAMEBUFFER_COMPLETE = 0x8CD5
VERTEX_SHADER, FRAGMENT_SHADER, GEOMETRY_SHADER = 0x8B31, 0x8B30, 0x8DD9
COMPILE_STATUS, LINK_STATUS, INFO_LOG_LENGTH = 0x8B81, 0x8B82, 0x8B84
ARRAY_BUFFER, ELEMENT_ARRAY_BUFFER, SHADER_STORAGE_BUFFER, UNIFORM_BUFFER = 0x8892, 0x8893, 0x90D2, 0x8A11
STATIC_DRAW, DYNAMIC_COPY, DYNAMIC_DRAW = 0x88E4, 0x88EA, 0x88E8
TRIANGLES, POINTS, TRIANGLE_STRIP = 0x0004, 0x0000, 0x0005
RASTERIZER_DISCARD = 0x8C89
WRITE_ONLY, READ_WRITE = 0x88B9, 0x88BA
ALL_BARRIER_BITS = 0xFFFFFFFF
SHADER_INCLUDE_ARB = 0x8DAE
TEXTURE0 = 0x84C0
DEPTH_TEST, CULL_FACE, BLEND = 0x0B71, 0x0B44, 0x0BE2
ALWAYS, LESS = 0x0207, 0x0201
MIN, FUNC_ADD = 0x8007, 0x8006
COLOR_BUFFER_BIT, DEPTH_BUFFER_BIT = 0x4000, 0x0100
MODELVIEW, PROJECTION = 0x1700, 0x1701
NO_ERROR = 0
PACK_ALIGNMENT, UNPACK_ALIGNMENT = 0x0D05, 0x0CF5
VERSION, RENDERER, SHADING_LANGUAGE_VERSION = 0x1F02, 0x1F01, 0x8B8C
FRONT, BACK = 0x0404, 0x0405

_SIGS = {
    "glGetString": (C.c_char_p, [C.c_uint]),
    "glGetError": (C.c_uint, []),
    "glGetIntegerv": (None, [C.c_uint, C.c_void_p]),
    "glGenTextures": (None, [C.c_int, C.c_void_p]),
    "glDeleteTextures": (None, [C.c_int, C.c_void_p]),
    "glBindTexture": (None, [C.c_uint, C.c_uint]),
    "glActiveTexture": (None, [C.c_uint]),
    "glTexParameteri": (None, [C.c_uint, C.c_uint, C.c_int]),
    "glTexImage2D": (None, [C.c_uint, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint, C.c_uint, C.c_void_p]),
    "glTexImage3D": (None, [C.c_uint, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint, C.c_uint, C.c_void_p]),
    "glTexSubImage2D": (None, [C.c_uint, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint, C.c_uint, C.c_void_p]),
    "glCompressedTexImage3D": (None, [C.c_uint, C.c_int, C.c_uint, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "glGetTexImage": (None, [C.c_uint, C.c_int, C.c_uint, C.c_uint, C.c_void_p]),
    "glClearTexImage": (None, [C.c_uint, C.c_int, C.c_uint, C.c_uint, C.c_void_p]),
    "glBindImageTexture": (None, [C.c_uint, C.c_uint, C.c_int, C.c_ubyte, C.c_int, C.c_uint, C.c_uint]),
    "glPixelStorei": (None, [C.c_uint, C.c_int]),
    "glGenFramebuffers": (None, [C.c_int, C.c_void_p]),
    "glBindFramebuffer": (None, [C.c_uint, C.c_uint]),
    "glFramebufferTextureLayer": (None, [C.c_uint, C.c_uint, C.c_uint, C.c_int, C.c_int]),
    "glFramebufferTexture2D": (None, [C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_int]),
    "glCheckFramebufferStatus": (C.c_uint, [C.c_uint]),
    "glDrawBuffers": (None, [C.c_int, C.c_void_p]),
    "glViewport": (None, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "glClearColor": (None, [C.c_float] * 4),
    "glClearDepth": (None, [C.c_double]),
    "glClear": (None, [C.c_uint]),
    "glEnable": (None, [C.c_uint]),
    "glDisable": (None, [C.c_uint]),
    "glDepthFunc": (None, [C.c_uint]),
    "glBlendEquation": (None, [C.c_uint]),
    "glCreateShader": (C.c_uint, [C.c_uint]),
    "glShaderSource": (None, [C.c_uint, C.c_int, C.c_void_p, C.c_void_p]),
    "glCompileShader": (None, [C.c_uint]),
    "glCompileShaderIncludeARB": (None, [C.c_uint, C.c_int, C.c_void_p, C.c_void_p]),
    "glNamedStringARB": (None, [C.c_uint, C.c_int, C.c_char_p, C.c_int, C.c_char_p]),
    "glGetShaderiv": (None, [C.c_uint, C.c_uint, C.c_void_p]),
    "glGetShaderInfoLog": (None, [C.c_uint, C.c_int, C.c_void_p, C.c_char_p]),
    "glCreateProgram": (C.c_uint, []),
    "glAttachShader": (None, [C.c_uint, C.c_uint]),
    "glBindAttribLocation": (None, [C.c_uint, C.c_uint, C.c_char_p]),
    "glLinkProgram": (None, [C.c_uint]),
    "glGetProgramiv": (None, [C.c_uint, C.c_uint, C.c_void_p]),
    "glGetProgramInfoLog": (None, [C.c_uint, C.c_int, C.c_void_p, C.c_char_p]),
    "glUseProgram": (None, [C.c_uint]),
    "glGetUniformLocation": (C.c_int, [C.c_uint, C.c_char_p]),
    "glGetAttribLocation": (C.c_int, [C.c_uint, C.c_char_p]),
    "glUniform1i": (None, [C.c_int, C.c_int]),
    "glUniform1ui": (None, [C.c_int, C.c_uint]),
    "glUniform1f": (None, [C.c_int, C.c_float]),
    "glUniform1iv": (None, [C.c_int, C.c_int, C.c_void_p]),
    "glUniform2fv": (None, [C.c_int, C.c_int, C.c_void_p]),
    "glUniform3fv": (None, [C.c_int, C.c_int, C.c_void_p]),
    "glUniform2uiv": (None, [C.c_int, C.c_int, C.c_void_p]),
    "glUniform3uiv": (None, [C.c_int, C.c_int, C.c_void_p]),
    "glUniformMatrix4fv": (None, [C.c_int, C.c_int, C.c_ubyte, C.c_void_p]),
    "glGenBuffers": (None, [C.c_int, C.c_void_p]),
    "glBindBuffer": (None, [C.c_uint, C.c_uint]),
    "glBufferData": (None, [C.c_uint, C.c_ssize_t, C.c_void_p, C.c_uint]),
    "glBufferSubData": (None, [C.c_uint, C.c_ssize_t, C.c_ssize_t, C.c_void_p]),
    "glGetBufferSubData": (None, [C.c_uint, C.c_ssize_t, C.c_ssize_t, C.c_void_p]),
    "glBindBufferRange": (None, [C.c_uint, C.c_uint, C.c_uint, C.c_ssize_t, C.c_ssize_t]),
    "glBindBufferBase": (None, [C.c_uint, C.c_uint, C.c_uint]),
    "glGenVertexArrays": (None, [C.c_int, C.c_void_p]),
    "glBindVertexArray": (None, [C.c_uint]),
    "glEnableVertexAttribArray": (None, [C.c_uint]),
    "glVertexAttribPointer": (None, [C.c_uint, C.c_int, C.c_uint, C.c_ubyte, C.c_int, C.c_void_p]),
    "glDrawArrays": (None, [C.c_uint, C.c_int, C.c_int]),
    "glDrawArraysInstanced": (None, [C.c_uint, C.c_int, C.c_int, C.c_int]),
    "glDrawElements": (None, [C.c_uint, C.c_int, C.c_uint, C.c_void_p]),
    "glDrawElementsInstanced": (None, [C.c_uint, C.c_int, C.c_uint, C.c_void_p, C.c_int]),
    "glMemoryBarrier": (None, [C.c_uint]),
    "glFinish": (None, []),
    "glMatrixMode": (None, [C.c_uint]),
    "glLoadMatrixf": (None, [C.c_void_p]),
    "glColorMask": (None, [C.c_ubyte] * 4),
    "glDepthMask": (None, [C.c_ubyte]),
    "glReadBuffer": (None, [C.c_uint]),
    "glReadPixels": (None, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint, C.c_uint, C.c_void_p]),
}


class GL:
    """gl = GL(); gl.glViewport(...).  Every call is followed by a glGetError check (this is a test harness)."""

    def __init__(self, compat=True, version=(4, 4)):
        for k, v in MESA_ENV.items():
            os.environ.setdefault(k, v)
        self._lib = C.CDLL(LIB_PATH)
        self._lib.glctx_proc.restype = C.c_void_p
        self._lib.glctx_proc.argtypes = [C.c_char_p]
        rc = self._lib.glctx_create(1 if compat else 0, version[0], version[1])
        if rc != 0:
            raise RuntimeError("glctx_create failed: %d" % rc)
        self._fn = {}
        self._get_error = self._raw("glGetError")

    def _raw(self, name):
        res, args = _SIGS[name]
        addr = self._lib.glctx_proc(name.encode())
        if not addr:
            raise RuntimeError("GL entry point %s is missing" % name)
        return C.CFUNCTYPE(res, *args)(addr)

    def __getattr__(self, name):
        if not name.startswith("gl"):
            raise AttributeError(name)
        fn = self._fn.get(name)
        if fn is None:
            raw = self._raw(name)

            def fn(*a, _raw=raw, _name=name):
                r = _raw(*a)
                e = self._get_error()
                if e != NO_ERROR:
                    raise RuntimeError("%s -> GL error 0x%04x" % (_name, e))
                return r
            self._fn[name] = fn
        return fn

    def info(self):
        return {k: self.glGetString(v).decode() for k, v in (("version", VERSION), ("renderer", RENDERER), ("glsl", SHADING_LANGUAGE_VERSION))}

    # ---- objects -----------------------------------------------------------------------------------------------
    def gen(self, what):
        n = C.c_uint(0)
        getattr(self, "glGen" + what)(1, C.byref(n))
        return n.value

    def named_string(self, name, text):
        b = text.encode()
        self.glNamedStringARB(SHADER_INCLUDE_ARB, len(name), name.encode(), len(b), b)

    def shader(self, kind, text, label):
        s = self.glCreateShader(kind)
        b = text.encode()
        src = C.c_char_p(b)
        n = C.c_int(len(b))
        self.glShaderSource(s, 1, C.byref(src), C.byref(n))
        self.glCompileShaderIncludeARB(s, 0, None, None)      # globjects' Shader::compile, no include paths: `#include </name>`
        ok = C.c_int(0)
        self.glGetShaderiv(s, COMPILE_STATUS, C.byref(ok))
        log = C.create_string_buffer(1 << 16)
        self.glGetShaderInfoLog(s, len(log), None, log)
        if not ok.value:
            raise RuntimeError("%s does not compile on Mesa:\n%s" % (label, log.value.decode()))
        return s

    def program(self, shaders, label, attribs=None):
        p = self.glCreateProgram()
        for s in shaders:
            self.glAttachShader(p, s)
        for loc, name in (attribs or {}).items():
            self.glBindAttribLocation(p, loc, name.encode())
        self.glLinkProgram(p)
        ok = C.c_int(0)
        self.glGetProgramiv(p, LINK_STATUS, C.byref(ok))
        log = C.create_string_buffer(1 << 16)
        self.glGetProgramInfoLog(p, len(log), None, log)
        if not ok.value:
            raise RuntimeError("%s does not link on Mesa:\n%s" % (label, log.value.decode()))
        return p

    def loc(self, prog, name, required=False):
        l = self.glGetUniformLocation(prog, name.encode())
        if l < 0 and required:
            raise RuntimeError("uniform %s is not active" % name)
        return l

    def texture(self, target, filt=LINEAR, wrap=CLAMP_TO_EDGE):
        """globjects Texture::createDefault: LINEAR min/mag, CLAMP_TO_EDGE s/t/r (globjects-0.5.0 Texture.cpp)"""
        t = self.gen("Textures")
        self.glActiveTexture(TEXTURE0)          # unit 0 is this harness's scratch unit: the reference binds nothing to it
        self.glBindTexture(target, t)
        self.glTexParameteri(target, TEXTURE_MIN_FILTER, filt)
        self.glTexParameteri(target, TEXTURE_MAG_FILTER, filt)
        self.glTexParameteri(target, TEXTURE_WRAP_S, wrap)
        self.glTexParameteri(target, TEXTURE_WRAP_T, wrap)
        self.glTexParameteri(target, TEXTURE_WRAP_R, wrap)
        return t

    def read_texture(self, target, tex, fmt, shape, dtype=np.float32, gltype=FLOAT):
        out = np.empty(shape, dtype)
        self.glPixelStorei(PACK_ALIGNMENT, 1)
        self.glActiveTexture(TEXTURE0)
        self.glBindTexture(target, tex)
        self.glGetTexImage(target, 0, fmt, gltype, out.ctypes.data)
        return out
