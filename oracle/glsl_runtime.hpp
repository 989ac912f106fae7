// glsl_runtime.hpp -- TEST INFRASTRUCTURE (build container only).  The C++ environment in which
// oracle/build_shader_ref.py compiles the TEXT of the reference's shaders on the hot path
// (glsl/pre_*.fs, glsl/inc_*.glsl, glsl/tsdf_integration.vs), read where it lies under /root/reference, into
// oracle/_ref/libref_shaders.so.  Vector types, operators and swizzles come from the reference's vendored
// external/glm-0.9.5.3 (GLM_SWIZZLE); everything the GLSL language leaves to the OpenGL driver is supplied here and is a
// STAND-IN, not reference code:
//   * texture() / texelFetch-free sampling: bound to the oracle's sampling functions (orc_tex3d_linear,
//     orc_tex2d_linear, orc_tex2d_linear_rgb8, orc_axis_nearest of oracle/rgbdr_oracle.c), with the filter state
//     the reference sets per texture;
//   * the built-ins whose results the GLSL spec does not fix bit for bit (pow, normalize, length, distance) in the
//     conventions DESIGN.md section 2 states for the oracle; the rest (abs, min, max, floor, sign, clamp, dot, cross)
//     are exact in any implementation;
//   * atomicAdd on the brick SSBO, imageStore on the TSDF image.
// So this pins nothing by the grading rule (the sampler is not the reference's): what it buys is that every
// arithmetic statement between two fetches is the reference's own text, compiled, not re-read.
#pragma once
#define GLM_SWIZZLE
#define GLM_FORCE_RADIANS
#include <glm/glm.hpp>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <array>
#include <map>
#include <string>
#include <vector>

extern "C" {
// oracle/rgbdr_oracle.c (librgbdr_oracle.so)
void orc_tex3d_linear(const float* vol, int ch, int rx, int ry, int rz, float u, float v, float w, float* out);
void orc_tex2d_linear(const float* img, int ch, int W, int H, float u, float v, float* out);
void orc_tex2d_linear_rgb8(const uint8_t* img, int W, int H, float u, float v, float* out);
int orc_axis_nearest(float s, int n);
void orc_tex2d_linear_mirrored_rgba(const float* col, int FW, int H, float u, float v, float* out);
}

namespace glslrt {
using glm::ivec2;
using glm::ivec3;
using glm::uvec2;
using glm::uvec3;
using glm::vec2;
using glm::vec3;
using glm::vec4;
using glm::mat4;
typedef unsigned int uint;

// ---- fragment state the fixed-function pipeline owns ------------------------------------------------------------
static vec4 gl_FragCoord;
static float gl_FragDepth;
static bool g_discarded;
struct DepthRange {
  float near = 0.0f, far = 1.0f;  // glDepthRange is never changed by the reference
};
static DepthRange gl_DepthRange;
#define discard          \
  do {                   \
    g_discarded = true;  \
    return;              \
  } while (0)

// ---- samplers ----------------------------------------------------------------------------------------------
struct sampler2DArray {  // [layers][H][W][ch] f32, or RGB8 when u8 is set; LINEAR or NEAREST, CLAMP_TO_EDGE
  const float* f32 = nullptr;
  const uint8_t* u8 = nullptr;
  int W = 0, H = 0, layers = 0, ch = 0;
  // 0 NEAREST; 1 LINEAR; 2 LINEAR state on a texture the shaders only ever sample at texel centres (m_textures_normal
  // in pre_quality.fs, m_textures_color in pre_boundary.fs / pre_quality.fs): the weight (px + .5) / W * W - .5 - px is
  // zero up to float rounding, texture units quantise it to 8 bits, and the oracle takes the
  // texel.  This mode does the same and COUNTS every lookup that is not within 2^-12 of a texel centre.
  int linear = 0;
};
static size_t g_offcentre_lookups = 0;
inline int centre_index(float s, int n)
{
  const float t = s * (float)n - 0.5f;
  const float r = floorf(t + 0.5f);
  if (!(fabsf(t - r) <= 1.0f / 4096.0f)) ++g_offcentre_lookups;
  const int i = (int)fminf(fmaxf(r, -1.0f), (float)n);
  return i < 0 ? 0 : (i > n - 1 ? n - 1 : i);
}
struct sampler3D {  // [rz][ry][rx][ch] f32, LINEAR, CLAMP_TO_EDGE
  const float* f32 = nullptr;
  int rx = 0, ry = 0, rz = 0, ch = 0;
};
struct sampler2D {  // [H][W][ch] f32; mode 0 NEAREST / CLAMP_TO_EDGE, 1 LINEAR / MIRRORED_REPEAT (the ViewLod colour atlas)
  const float* f32 = nullptr;
  int W = 0, H = 0, ch = 0;
  int mode = 0;
};
struct image2D {  // r32f, write only
  float* f32 = nullptr;
  int W = 0, H = 0;
};
struct image3D {  // r32f, write only: x fastest
  float* f32 = nullptr;
  int X = 0, Y = 0, Z = 0;
};

inline vec4 texture(const sampler2DArray& s, const vec3& c)
{
  // array layer = clamp(round(layer), 0, layers - 1), never filtered (GL 4.4 section 8.9.3)
  int layer = (int)floorf(c.z + 0.5f);
  layer = layer < 0 ? 0 : (layer > s.layers - 1 ? s.layers - 1 : layer);
  float out[4] = {0.0f, 0.0f, 0.0f, 1.0f};
  const size_t px = (size_t)s.W * s.H;
  if (s.u8) {
    orc_tex2d_linear_rgb8(s.u8 + px * 3 * layer, s.W, s.H, c.x, c.y, out);
  } else if (s.linear == 1) {
    orc_tex2d_linear(s.f32 + px * s.ch * layer, s.ch, s.W, s.H, c.x, c.y, out);
  } else if (s.linear == 2) {
    const int ix = centre_index(c.x, s.W), iy = centre_index(c.y, s.H);
    const float* t = s.f32 + (px * layer + (size_t)iy * s.W + ix) * s.ch;
    for (int k = 0; k < s.ch; ++k) out[k] = t[k];
  } else {
    const int ix = orc_axis_nearest(c.x, s.W), iy = orc_axis_nearest(c.y, s.H);
    const float* t = s.f32 + (px * layer + (size_t)iy * s.W + ix) * s.ch;
    for (int k = 0; k < s.ch; ++k) out[k] = t[k];
  }
  return vec4(out[0], out[1], out[2], out[3]);
}
inline vec4 texture(const sampler3D& s, const vec3& c)
{
  float out[4] = {0.0f, 0.0f, 0.0f, 1.0f};
  orc_tex3d_linear(s.f32, s.ch, s.rx, s.ry, s.rz, c.x, c.y, c.z, out);
  return vec4(out[0], out[1], out[2], out[3]);
}
// texelFetch outside the texture is undefined in GL; the oracle's convention (zeros) -- tsdf_inpaint.fs reaches
// outside at the atlas borders
inline vec4 texelFetch(const sampler2D& s, const ivec2& p, int /*lod*/)
{
  float out[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  if (p.x >= 0 && p.y >= 0 && p.x < s.W && p.y < s.H)
    for (int k = 0; k < s.ch; ++k) out[k] = s.f32[((size_t)p.y * s.W + p.x) * s.ch + k];
  return vec4(out[0], out[1], out[2], out[3]);
}
inline vec4 texture(const sampler2D& s, const vec2& c)
{
  float out[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  if (s.mode == 1 && s.ch == 4) {
    orc_tex2d_linear_mirrored_rgba(s.f32, s.W, s.H, c.x, c.y, out);
  } else {
    const int ix = orc_axis_nearest(c.x, s.W), iy = orc_axis_nearest(c.y, s.H);
    for (int k = 0; k < s.ch; ++k) out[k] = s.f32[((size_t)iy * s.W + ix) * s.ch + k];
  }
  return vec4(out[0], out[1], out[2], out[3]);
}
inline void imageStore(image2D& img, const ivec2& p, const vec4& v)
{
  if (p.x < 0 || p.y < 0 || p.x >= img.W || p.y >= img.H) return;
  img.f32[(size_t)p.y * img.W + p.x] = v.x;
}
inline void imageStore(image3D& img, const ivec3& p, const vec4& v)
{
  if (p.x < 0 || p.y < 0 || p.z < 0 || p.x >= img.X || p.y >= img.Y || p.z >= img.Z) return;  // GL: out-of-bounds stores are dropped
  img.f32[((size_t)p.z * img.Y + p.y) * img.X + p.x] = v.x;
}

// ---- the SSBO of inc_bricks.glsl: an unsized uint array.  Indices outside the buffer (the shader converts a
// negative float to uvec3 for positions outside the brick grid) land in a scratch word and are counted, so that a
// scene which produces any shows up instead of corrupting memory.
struct uint_buffer {
  uint* data = nullptr;
  size_t n = 0;
  size_t out_of_range = 0;
  uint scratch = 0;
  uint& operator[](uint i)
  {
    if ((size_t)i < n) return data[i];
    ++out_of_range;
    return scratch;
  }
};
inline uint atomicAdd(uint& mem, uint v)
{
  const uint old = mem;
  mem += v;
  return old;
}

// ---- built-ins -----------------------------------------------------------------------------------------------
// exact in every implementation
inline float abs(float x) { return fabsf(x); }
inline vec3 abs(const vec3& v) { return vec3(fabsf(v.x), fabsf(v.y), fabsf(v.z)); }
inline float min(float a, float b) { return b < a ? b : a; }  // GLSL: y < x ? y : x
inline float max(float a, float b) { return a < b ? b : a;  } // GLSL: x < y ? y : x
inline float floor(float x) { return floorf(x); }
inline vec3 floor(const vec3& v) { return vec3(floorf(v.x), floorf(v.y), floorf(v.z)); }
inline float sign1(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }
inline vec3 sign(const vec3& v) { return vec3(sign1(v.x), sign1(v.y), sign1(v.z)); }
inline ivec3 clamp(const ivec3& v, const ivec3& lo, const ivec3& hi)
{
  auto c = [](int x, int a, int b) { return x < a ? a : (x > b ? b : x); };  // min(max(x, lo), hi)
  return ivec3(c(v.x, lo.x, hi.x), c(v.y, lo.y, hi.y), c(v.z, lo.z, hi.z));
}
// association as in glm (and in the oracle): (x*x + y*y) + z*z
inline float dot(const vec3& a, const vec3& b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline float dot(const vec2& a, const vec2& b) { return a.x * b.x + a.y * b.y; }
inline vec3 cross(const vec3& x, const vec3& y)
{
  return vec3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
inline float ceil(float x) { return ceilf(x); }
inline vec2 min(const vec2& a, const vec2& b) { return vec2(min(a.x, b.x), min(a.y, b.y)); }
inline vec2 max(const vec2& a, const vec2& b) { return vec2(max(a.x, b.x), max(a.y, b.y)); }
inline vec3 min(const vec3& a, const vec3& b) { return vec3(min(a.x, b.x), min(a.y, b.y), min(a.z, b.z)); }
inline vec3 max(const vec3& a, const vec3& b) { return vec3(max(a.x, b.x), max(a.y, b.y), max(a.z, b.z)); }
inline vec2 floor(const vec2& v) { return vec2(floorf(v.x), floorf(v.y)); }
// clamp(x, lo, hi) = min(max(x, lo), hi), also where lo > hi (tsdf_colorfill.fs:33 beyond the last LOD; GL: undefined)
inline vec2 clamp(const vec2& v, const vec2& lo, const vec2& hi) { return min(max(v, lo), hi); }
// inverse(mat4) is evaluated by the GPU in the reference (tsdf_raymarch.fs:387-388); the oracle takes the inverses the
// host computed (rgbdr_view: modelview_inv, vol_to_world_inv).  The harness registers those pairs here.
struct InversePair {
  mat4 m, inv;
};
static std::vector<InversePair> g_inverses;
inline mat4 inverse(const mat4& m)
{
  for (const InversePair& p : g_inverses)
    if (std::memcmp(&p.m, &m, sizeof(mat4)) == 0) return p.inv;
  return glm::inverse(m);
}
// driver-defined in GLSL; the oracle's conventions (DESIGN.md section 2 "Numeric conventions")
inline float length(const vec2& v) { return sqrtf(dot(v, v)); }
inline float length(const vec3& v) { return sqrtf(dot(v, v)); }
inline float distance(float a, float b) { return fabsf(a - b); }
inline float distance(const vec3& a, const vec3& b) { return length(vec3(a.x - b.x, a.y - b.y, a.z - b.z)); }
inline float distance(const vec2& a, const vec2& b) { return length(vec2(a.x - b.x, a.y - b.y)); }
inline vec3 normalize(const vec3& v)
{
  const float l = sqrtf(dot(v, v));
  return vec3(v.x / l, v.y / l, v.z / l);
}
inline float pow(float x, float y)
{
  if (y == 2.0f) return x * x;  // constant integer exponents are products (pre_quality.fs:109-114)
  if (y == 6.0f) {
    const float x2 = x * x, x4 = x2 * x2;
    return x4 * x2;
  }
  if (y == 20.0f) {  // shading.glsl:45
    const float r2 = x * x, r4 = r2 * r2, r8 = r4 * r4, r16 = r8 * r8;
    return r16 * r4;
  }
  return powf(x, y);
}

// GLSL converts integer vectors to float vectors implicitly (tsdf_integration.vs:57 `position * res_tsdf`,
// framebuffer_transfer.fs:14 `pass_TexCoord * resolution_tex`, tsdf_colorfill.fs:24 `ivec2(...) + vec2(...) * pos`);
// glm has no mixed operators
inline vec3 operator*(const vec3& a, const uvec3& b) { return vec3(a.x * (float)b.x, a.y * (float)b.y, a.z * (float)b.z); }
inline vec2 operator*(const vec2& a, const uvec2& b) { return vec2(a.x * (float)b.x, a.y * (float)b.y); }
inline vec2 operator+(const ivec2& a, const vec2& b) { return vec2((float)a.x + b.x, (float)a.y + b.y); }

// ---- registry: uniforms / samplers / outputs by name, filled by the generated code ---------------------------
struct Slot {
  void* ptr;
  size_t bytes;  // of one element
  int count;     // array length (1 for scalars)
  char kind;     // 'u' plain data, 's' sampler2DArray, 't' sampler3D, 'i' image, 'b' uint_buffer, 'o' output
};
typedef std::map<std::string, Slot> Registry;
}  // namespace glslrt

