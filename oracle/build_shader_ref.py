#!/usr/bin/env python3
"""TEST INFRASTRUCTURE, build container only: compiles the TEXT of the reference's hot-path shaders as C++.

Reads glsl/pre_morph.fs, pre_depth.fs, pre_boundary.fs, pre_normal.fs, pre_quality.fs, tsdf_integration.vs and the
files they #include (inc_bbox_test.glsl, inc_color.glsl, inc_bricks.glsl -- the names NetKinectArray.cpp:90,208-209
registers them under) WHERE THEY LIE under /root/reference, rewrites only what is GLSL *syntax* (see `transform`),
writes the result to a scratch directory OUTSIDE this repository, and compiles it with g++ against the reference's
vendored external/glm-0.9.5.3 and oracle/glsl_runtime.hpp into oracle/_ref/libref_shaders.so (git-ignored).  Nothing of
the reference's text is stored in the repository or in anything that is committed; the .so is a build product like
oracle/_ref/libref_shim.so.

What is rewritten, mechanically (every arithmetic statement, comparison, loop and constant stays as written):
  * `#version`, `#extension` lines removed; `#include </name>` expanded in place;
  * qualifiers: `uniform T x;` `noperspective in T x;` `in T x;` `layout(...) out T x;` become globals `T x;` that the
    harness reads / writes by name; `T[N] x` becomes `T x[N]`; interface blocks (`layout(std140) uniform BBox {...}`,
    `layout(std430) buffer Bricks {...}`) are opened into globals, an unsized `uint[] x` becomes a checked buffer;
  * parameter qualifiers `const in T`, `in T` dropped, `out T` becomes `T&`;
  * swizzles `.xy` `.rgb` ... become glm's swizzle calls `.xy()`; float literals get the `f` suffix GLSL gives them
    implicitly (`1.0` is a 32-bit float in GLSL, a double in C++);
  * `void main(void)` becomes `void shader_main()`.
usage: build_shader_ref.py [--ref /root/reference] [--out oracle/_ref/libref_shaders.so] [--keep-src DIR]
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
# `#include </x>` names -> files: globjects::NamedString::create calls of framework/NetKinectArray.cpp:90,208-209
# ... and reconstruction.cpp:24 ("/shading.glsl")
INCLUDES = {"/inc_bbox_test.glsl": "inc_bbox_test.glsl", "/inc_color.glsl": "inc_color.glsl", "/bricks.glsl": "inc_bricks.glsl",
            "/shading.glsl": "shading.glsl"}
SHADERS = {"pre_morph": "pre_morph.fs", "pre_depth": "pre_depth.fs", "pre_boundary": "pre_boundary.fs", "pre_normal": "pre_normal.fs",
           "pre_quality": "pre_quality.fs", "tsdf_integration": "tsdf_integration.vs",
           # consumers of the volume: the ray-marcher and the screen-space hole filling
           "tsdf_raymarch": "tsdf_raymarch.fs", "framebuffer_transfer": "framebuffer_transfer.fs", "tsdf_inpaint": "tsdf_inpaint.fs",
           "tsdf_colorfill": "tsdf_colorfill.fs"}
KIND = {"sampler2DArray": "s", "sampler3D": "t", "sampler2D": "x", "image3D": "i", "image2D": "i", "uint_buffer": "b"}
SCALARS = r"(?:float|int|uint|bool|vec[234]|ivec[234]|uvec[234]|mat4)"
FLOAT_LIT = re.compile(r"(?<![\w.])(\d+\.\d*|\.\d+)([eE][+-]?\d+)?(?![\w.])")
SWIZZLE = re.compile(r"\.([xyzw]{2,4}|[rgba]{2,4}|[stpq]{2,4})\b(?!\s*\()")


def expand_includes(text, glsl_dir):
    def repl(m):
        return expand_includes(open(os.path.join(glsl_dir, INCLUDES[m.group(1)])).read(), glsl_dir)
    return re.sub(r"^[ \t]*#include\s*<([^>]+)>[ \t]*$", repl, text, flags=re.M)


def transform(text):
    """-> (C++ text, [(name, type, count, role)]) ; role: u uniform, i input, o output"""
    slots = []
    out = []
    in_block = False
    for line in text.splitlines():
        s = line.strip()
        if s.startswith("#version") or s.startswith("#extension"):
            continue
        code = line.split("//")[0]
        # interface blocks
        m = re.match(r"\s*layout\s*\([^)]*\)\s*(uniform|buffer)\s+\w+\s*\{\s*$", code)
        if m:
            in_block = True
            continue
        if in_block:
            if re.match(r"\s*\}\s*;\s*$", code):
                in_block = False
                continue
            m = re.match(r"\s*(\w+)\s*\[\s*\]\s*(\w+)\s*;", code)          # uint[] bricks;
            if m:
                assert m.group(1) == "uint"
                out.append("uint_buffer %s;" % m.group(2))
                slots.append((m.group(2), "uint_buffer", 1, "u"))
                continue
            m = re.match(r"\s*(\w+)\s+(\w+)\s*(?:\[\s*(\d+)\s*\])?\s*;", code)
            if m:
                cnt = int(m.group(3) or 1)
                out.append("%s %s%s;" % (m.group(1), m.group(2), "[%d]" % cnt if m.group(3) else ""))
                slots.append((m.group(2), m.group(1), cnt, "u"))
                continue
            if not code.strip():
                continue
            raise SystemExit("interface block member not understood: %r" % line)
        # globals with storage qualifiers
        m = re.match(r"\s*(?:layout\s*\([^)]*\)\s*)?(?:noperspective\s+)?(uniform|in|out)\s+(\w+)\s*(?:\[\s*(\d+)\s*\])?\s+(\w+)\s*;", code)
        if m and m.group(4).startswith("gl_") and m.group(1) != "uniform":
            continue                                  # gl_FragCoord / gl_FragDepth belong to the runtime
        if m:
            role = {"uniform": "u", "in": "i", "out": "o"}[m.group(1)]
            cnt = int(m.group(3) or 1)
            out.append("%s %s%s;" % (m.group(2), m.group(4), "[%d]" % cnt if m.group(3) else ""))
            slots.append((m.group(4), m.group(2), cnt, role))
            continue
        if re.match(r"\s*out\s+float\s+gl_FragDepth\s*;", code):
            continue
        out.append(line)
    body = "\n".join(out)
    assert not re.search(r"\.([xyzw]{2,4}|[rgba]{2,4})\s*[-+*/]?=[^=]", body), "assignment to a swizzle: not supported"
    body = re.sub(r"\bvoid\s+main\s*\(\s*(void)?\s*\)", "void shader_main()", body)
    body = re.sub(r"\bconst\s+in\s+", "const ", body)
    body = re.sub(r"([(,]\s*)in\s+(?=\w+\s+\w+\s*[,)])", r"\1", body)
    body = re.sub(r"([(,]\s*)out\s+(\w+)\s+(?=\w+\s*[,)])", r"\1\2& ", body)
    body = SWIZZLE.sub(lambda m: ".%s()" % m.group(1), body)
    body = rewrite_arrays(body)
    body, inits = hoist_global_initialisers(body)
    # float literals: outside of preprocessor lines and comments
    lines = []
    for line in body.splitlines():
        if line.lstrip().startswith("#"):
            lines.append(line)
            continue
        parts = line.split("//", 1)
        parts[0] = FLOAT_LIT.sub(lambda m: m.group(0) + "f", parts[0])
        lines.append("//".join(parts))
    return "\n".join(lines) + "\nstatic void shader_globals_init()\n{\n" + "".join(
        "  " + FLOAT_LIT.sub(lambda m: m.group(0) + "f", i) + "\n" for i in inits) + "}\n", slots


def balanced(text, open_at):
    """index just past the parenthesis that closes the one at text[open_at]"""
    depth = 0
    for i in range(open_at, len(text)):
        depth += text[i] == "("
        depth -= text[i] == ")"
        if depth == 0:
            return i + 1
    raise SystemExit("unbalanced parentheses")


def rewrite_arrays(body):
    """GLSL array syntax C++ does not have: `T name[N] = T[N](a, b, ...)` (array constructor), `T[N] name` (array-typed
    variable), `T[N] f(...)` (array-returning function) -> std::array<T, N>"""
    out, pos = [], 0
    for m in re.finditer(r"\b(const\s+)?(\w+)\s+(\w+)\s*\[\s*(\d+)\s*\]\s*=\s*(\w+)\s*\[\s*(\d+)\s*\]\s*\(", body):
        if m.start() < pos or m.group(2) != m.group(5) or m.group(4) != m.group(6):
            continue
        end = balanced(body, m.end() - 1)
        out.append(body[pos:m.start()])
        out.append("%sstd::array<%s, %s> %s = {{%s}}" % (m.group(1) or "", m.group(2), m.group(4), m.group(3), body[m.end():end - 1]))
        pos = end
    out.append(body[pos:])
    body = "".join(out)
    return re.sub(r"\b(\w+)\s*\[\s*(\d+)\s*\]\s+(\w+)\b", lambda m: "std::array<%s, %s> %s" % m.groups(), body)


def hoist_global_initialisers(body):
    """A non-const global with an initialiser is initialised at the start of every shader invocation in GLSL (it may
    depend on uniforms: `float sampleDistance = limit * 0.5f;`); a C++ global would be initialised once at load time.
    Such declarations keep only `T name;` and the assignment moves into shader_globals_init(), run before main()."""
    lines, inits, depth = [], [], 0
    for line in body.splitlines():
        code = line.split("//")[0]
        m = re.match(r"^(\s*)(%s)\s+(\w+)\s*=\s*(.+);\s*$" % SCALARS, code) if depth == 0 else None
        if m:
            lines.append("%s%s %s;" % (m.group(1), m.group(2), m.group(3)))
            inits.append("%s = %s;" % (m.group(3), m.group(4)))
        else:
            lines.append(line)
        depth += code.count("{") - code.count("}")
    return "\n".join(lines), inits


HARNESS = r'''
// ---- generated harness (oracle/build_shader_ref.py) ----------------------------------------------------------
static Registry& registry()
{
  static Registry r;
  if (r.empty()) {
%(reg)s
  }
  return r;
}
static std::map<std::string, float*> g_out;
}  // namespace shader_%(name)s
}  // namespace glslrt

using namespace glslrt;
using namespace glslrt::shader_%(name)s;
extern "C" {
// copies `bytes` into element `index` of the named uniform / input (plain data), or binds a sampler / image / buffer
// from a descriptor {ptr, dims...} (see oracle/shader_ref.py)
__attribute__((visibility("default"))) int shref_%(name)s_set(const char* name, int index, const void* data, size_t bytes)
{
  auto it = registry().find(name);
  if (it == registry().end()) return -1;
  const Slot& s = it->second;
  if (index < 0 || index >= s.count || bytes != s.bytes) return -2;
  std::memcpy((char*)s.ptr + (size_t)index * s.bytes, data, bytes);
  return 0;
}
__attribute__((visibility("default"))) int shref_%(name)s_bind_out(const char* name, float* dst)
{
  auto it = registry().find(name);
  if (it == registry().end() || it->second.kind != 'o') return -1;
  g_out[name] = dst;
  return 0;
}
__attribute__((visibility("default"))) size_t shref_%(name)s_buffer_out_of_range(const char* name)
{
  auto it = registry().find(name);
  if (it == registry().end() || it->second.kind != 'b') return (size_t)-1;
  return ((uint_buffer*)it->second.ptr)->out_of_range;
}
__attribute__((visibility("default"))) size_t shref_%(name)s_offcentre_lookups() { return g_offcentre_lookups; }
static void store_outputs(size_t element)
{
  for (auto& o : g_out) {
    const Slot& s = registry()[o.first];
    std::memcpy(o.second + element * (s.bytes / sizeof(float)), s.ptr, s.bytes);
  }
}
%(run)s
}
'''

RUN_FRAGMENT = r'''
// one fragment per pixel of a W x H target: pass_TexCoord = the texel centre (screen_quad.cpp:11-15,
// texture_passthrough.vs), rows bottom-up like the reference's viewport; outputs are stored per pixel
__attribute__((visibility("default"))) void shref_%(name)s_run(int W, int H)
{
  for (int py = 0; py < H; ++py)
    for (int px = 0; px < W; ++px) {
      pass_TexCoord = vec2(((float)px + 0.5f) / (float)W, ((float)py + 0.5f) / (float)H);
      shader_globals_init();
      shader_main();
      store_outputs((size_t)py * W + px);
    }
}
'''
RUN_VERTEX = r'''
// one vertex per voxel centre, positions as VolumeSampler builds them (volume_sampler.cpp:13-22), z rows [z0, z1)
__attribute__((visibility("default"))) void shref_%(name)s_run(int X, int Y, int Z, int z0, int z1)
{
  const float stepX = 1.0f / X, stepY = 1.0f / Y, stepZ = 1.0f / Z;
  for (int z = z0; z < z1; ++z)
    for (int y = 0; y < Y; ++y)
      for (int x = 0; x < X; ++x) {
        in_Position = vec3((x + 0.5f) * stepX, (y + 0.5f) * stepY, (z + 0.5f) * stepZ);
        shader_globals_init();
        shader_main();
      }
}
'''


RUN_RAYMARCH = r'''
// tsdf_raymarch.fs over a W x H viewport.  Stand-ins for the fixed-function parts (as in the oracle): the unit cube is
// "rasterised" onto a pixel when the ray through its centre meets the cube in front of the camera (the shader's own
// intersectBox decides), pass_Position is the volume-space far-plane point of that ray (screenToVol with the host's
// inverses), the framebuffer starts cleared to (0,1,0,0) / depth 1 (ViewLod::enable), gl_FragDepth is clamped and
// tested GL_LESS.  inverse(mat4) inside the shader returns the host's inverses (registered pairs).
__attribute__((visibility("default"))) void shref_%(name)s_add_inverse(const float* m, const float* inv)
{
  InversePair p;
  std::memcpy(&p.m, m, sizeof(mat4));
  std::memcpy(&p.inv, inv, sizeof(mat4));
  g_inverses.push_back(p);
}
__attribute__((visibility("default"))) void shref_%(name)s_run(int W, int H, const float* modelview_inv, const float* vol_to_world_inv,
                                                               float* out_color, float* out_depth)
{
  mat4 mvi, v2wi;
  std::memcpy(&mvi, modelview_inv, sizeof(mat4));
  std::memcpy(&v2wi, vol_to_world_inv, sizeof(mat4));
  for (int py = 0; py < H; ++py)
    for (int px = 0; px < W; ++px) {
      const size_t o = (size_t)py * W + px;
      out_color[o * 4 + 0] = 0.0f;
      out_color[o * 4 + 1] = 1.0f;
      out_color[o * 4 + 2] = 0.0f;
      out_color[o * 4 + 3] = 0.0f;
      out_depth[o] = 1.0f;
      shader_globals_init();
      const vec4 pc = img_to_eye_curr * vec4((float)px + 0.5f, (float)py + 0.5f, 1.0f, 1.0f);
      const vec4 es = vec4(pc.x / pc.w, pc.y / pc.w, pc.z / pc.w, 1.0f);
      const vec4 tv = v2wi * (mvi * es);
      pass_Position = vec3(tv.x, tv.y, tv.z);
      float t0 = 0.0f, t1 = 0.0f;
      const bool is_t0 = intersectBox(CameraPos, normalize(pass_Position - CameraPos) * sampleDistance, t0, t1);
      if (!is_t0 || !(t1 > 0.0f)) continue;  // no fragment
      gl_FragCoord = vec4((float)px + 0.5f, (float)py + 0.5f, 0.0f, 1.0f);
      gl_FragDepth = 0.0f;
      g_discarded = false;
      shader_main();
      if (g_discarded) continue;
      const float d = gl_FragDepth < 0.0f ? 0.0f : (gl_FragDepth > 1.0f ? 1.0f : gl_FragDepth);
      if (!(d < 1.0f)) continue;
      std::memcpy(out_color + o * 4, &out_Color, 16);
      out_depth[o] = d;
    }
}
'''
RUN_VIEWPORT = r'''
// a ScreenQuad::draw into the viewport (x0, y0, w, h) of an FW-wide RGBA32F + depth atlas (ViewLod::enable,
// view_lod.cpp:73-92) with glDepthFunc(GL_ALWAYS) (recon_integration.cpp:281): every fragment writes out_FragColor and
// gl_FragDepth.  pass_TexCoord runs over the quad, gl_FragCoord is the window pixel (integer centres where the shader
// declares layout(pixel_center_integer)).
__attribute__((visibility("default"))) void shref_%(name)s_run(int x0, int y0, int w, int h, int integer_centres, float* color, float* depth,
                                                               int FW)
{
  for (int py = 0; py < h; ++py)
    for (int px = 0; px < w; ++px) {
      const float c = integer_centres ? 0.0f : 0.5f;
      gl_FragCoord = vec4((float)(x0 + px) + c, (float)(y0 + py) + c, 0.0f, 1.0f);
      pass_TexCoord = vec2(((float)px + 0.5f) / (float)w, ((float)py + 0.5f) / (float)h);
      gl_FragDepth = 0.0f;
      g_discarded = false;
      shader_globals_init();
      shader_main();
      const size_t o = (size_t)(y0 + py) * FW + (x0 + px);
      std::memcpy(color + o * 4, &out_FragColor, 16);
      depth[o] = gl_FragDepth;
    }
}
'''


def generate(name, glsl_dir):
    text = expand_includes(open(os.path.join(glsl_dir, SHADERS[name])).read(), glsl_dir)
    body, slots = transform(text)
    reg = []
    for (n, t, cnt, role) in slots:
        kind = KIND.get(t, "o" if role == "o" else "u")
        reg.append('    r["%s"] = Slot{(void*)&%s, sizeof(%s), %d, \'%s\'};' % (n, n + ("[0]" if cnt > 1 else ""), t, cnt, kind))
    run = {"tsdf_integration": RUN_VERTEX, "tsdf_raymarch": RUN_RAYMARCH, "framebuffer_transfer": RUN_VIEWPORT,
           "tsdf_inpaint": RUN_VIEWPORT, "tsdf_colorfill": RUN_VIEWPORT}.get(name, RUN_FRAGMENT) % {"name": name}
    return ('// GENERATED from the reference\'s glsl/%s by oracle/build_shader_ref.py -- scratch file, never committed\n'
            '#include "glsl_runtime.hpp"\nnamespace glslrt {\nnamespace shader_%s {\n' % (SHADERS[name], name)
            + body + "\n" + HARNESS % {"name": name, "reg": "\n".join(reg), "run": run})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(HERE, "_ref", "libref_shaders.so"))
    ap.add_argument("--keep-src", default="", help="scratch directory to keep the generated C++ in (must be outside the repository)")
    args = ap.parse_args()
    glsl_dir = os.path.join(args.ref, "glsl")
    if not os.path.isdir(glsl_dir):
        print("reference checkout absent: oracle/_ref/libref_shaders.so not rebuilt")
        return 0
    root = os.path.dirname(HERE)
    scratch = args.keep_src or tempfile.mkdtemp(prefix="rgbdr_shader_ref_")
    if os.path.abspath(scratch).startswith(os.path.abspath(root) + os.sep):
        raise SystemExit("the generated sources carry the reference's text: keep them outside the repository")
    os.makedirs(scratch, exist_ok=True)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    srcs = []
    for name in SHADERS:
        p = os.path.join(scratch, "shader_%s.cpp" % name)
        with open(p, "w") as f:
            f.write(generate(name, glsl_dir))
        srcs.append(p)
    cmd = ["g++", "-O1", "-std=c++14", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-fvisibility=hidden", "-w",
           "-I", HERE, "-I", os.path.join(args.ref, "external", "glm-0.9.5.3"), "-o", args.out] + srcs + \
          ["-L", HERE, "-lrgbdr_oracle", "-Wl,-rpath,$ORIGIN/..", "-lm"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stderr[-6000:])
        print("oracle/_ref: compiling the reference's shader text FAILED (generated sources: %s)" % scratch)
        return 1
    if not args.keep_src:
        for p in srcs:
            os.remove(p)
        os.rmdir(scratch)
    print("built oracle/_ref/libref_shaders.so from the shader text under %s" % glsl_dir)
    return 0


if __name__ == "__main__":
    sys.exit(main())
