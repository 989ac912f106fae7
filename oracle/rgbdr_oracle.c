/*
 * rgbdr_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar CPU restatement of the rgbd-recon depth-preprocessing + TSDF-integration
 * hot path (the GLSL programs the reference runs on an OpenGL 4.4 device).  It is
 * the checker for the HIP kernels: only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this file's library.  The product library
 * (rgbd-recon_amd/csrc) never links, loads or calls anything in oracle/.
 *
 * PARITY PINNING.  The reference ships no tests, golden vectors or recordings
 * (SURVEY.md section 4).  Since round 3 the pass arithmetic below is pinned against
 * OUTPUTS OF THE REFERENCE'S OWN SHADERS RUN IN THE BUILD CONTAINER: the GLSL files where
 * they lie under /root/reference/glsl are compiled by Mesa's GLSL compiler and executed by
 * llvmpipe (Mesa 23.2.1, OpenGL 4.5 compatibility profile, reached without an X server
 * through the image's own swrast_dri.so: oracle/gl_context.c, oracle/gl_ref.py), through
 * one frame in the reference's host order -- pre_morph, pre_depth, pre_boundary, pre_normal
 * (+ mark_brick's SSBO atomics), pre_quality, tsdf_integration.vs (points + imageStore),
 * bricks.{vs,gs,fs} (rasteriser + MIN blending), tsdf_raymarch.{vs,fs}, framebuffer_transfer /
 * tsdf_inpaint / tsdf_colorfill -- and frozen as tests/golden/gl_passes_*.npz / gl_views_*.npz.
 * tests/test_gl_ref.py holds this file to them (and the HIP path, on the GPU box): depth,
 * filtered depth, boundary classes, silhouettes and brick counters bit for bit; normals,
 * quality, Lab and the TSDF within the ulp-level tolerances stated there (TSDF: <= 2e-9
 * observed, 1e-7 allowed, no voxel changes class); peels, ray-march and hole filling per
 * pixel.  What that run does NOT pin: the reference's HOST code is restated by gl_ref.py
 * (texture formats / filters / bindings / uniforms, read from NetKinectArray.cpp,
 * recon_integration.cpp, CalibVolumes.cpp), and llvmpipe is not the NVIDIA driver the
 * reference was developed on (filter weights are float, not 8-bit fixed point; pow / exp /
 * inversesqrt are llvmpipe's approximations; pow(x < 0, y) is NaN there).  Two in-memory
 * edits of the text and the one vertex-stage defect of llvmpipe the harness pads around
 * are listed at the top of oracle/gl_ref.py.
 * Also pinned, against reference C++ compiled from /root/reference (oracle/_ref, see
 * oracle/Makefile and oracle/ref_shim.cpp):
 *   - the calibration-volume file format (orc_lut_write / orc_lut_read versus
 *     framework/calibration/calibration_volume.hpp:30-79),
 *   - the record layouts (kinect::xyz 12 B, kinect::uv 8 B, framework/DataTypes.h:12-35),
 *   - trilinear interpolation versus kinect::getTrilinear
 *     (framework/DataTypes.cpp:115-163) to a few ulp (different but equivalent
 *     association of the lerps),
 *   - DXT1 / DXT5 colour decoding (orc_decode_dxt) bit for bit versus squish::DecompressImage
 *     (external/squish, the decoder of NetKinectArray.cpp:633).
 * The same shim also pins the C++ host mirror's sensor-yml scanner and .stream reader against
 * kinect::CalibrationFiles / sys::FileBuffer (tests/test_oracle_ref.py).
 * ALSO CHECKED (round 3, before the Mesa run existed): the TEXT of the reference's shaders -- pre_morph / pre_depth / pre_boundary /
 * pre_normal / pre_quality .fs, tsdf_integration.vs, tsdf_raymarch.fs + shading.glsl, framebuffer_transfer / tsdf_inpaint /
 * tsdf_colorfill .fs and their includes -- is compiled as C++ from /root/reference where it lies (oracle/build_shader_ref.py,
 * against the reference's vendored glm) and run against the functions below on synthetic scenes: bit-identical in every
 * image, counter, volume and frame (tests/test_shader_ref.py).  The texture samplers and the driver-defined built-ins of
 * that harness are stand-ins bound to THIS file's conventions (oracle/glsl_runtime.hpp), so it is not a run of the reference;
 * it shows that no statement of a shader was mis-read, bit for bit, where the Mesa run can only say "within llvmpipe's ulps".
 *
 * NUMERIC CONVENTIONS (decisions where GLSL / GL leave the result to the driver):
 *   - all arithmetic is IEEE-754 binary32, no FMA contraction (-ffp-contract=off),
 *     division and sqrt correctly rounded;
 *   - LINEAR filtering follows GL 4.4 section 8.14: t = s*n - 0.5, i0 = floor(t),
 *     a = t - i0, i1 = i0 + 1, both clamped to [0, n-1] (CLAMP_TO_EDGE), and the
 *     blend is lerp(T0, T1, a) = T0 + a*(T1 - T0), x first, then y, then z.  That
 *     form returns T0 exactly when T0 == T1 (as fixed-point texture hardware
 *     does), which matters for `silhouette < 1.0` in tsdf_integration.vs:33;
 *   - NEAREST: i = clamp(floor(s*n), 0, n-1);
 *   - a NaN coordinate selects texel 0 and yields a NaN blend weight;
 *   - fragment (px,py) of a WxH target has texcoord ((px+.5)/W, (py+.5)/H)
 *     (framework/rendering/screen_quad.cpp:7-35, glsl/texture_passthrough.vs:1-12);
 *     tap offsets k*texSizeInv on NEAREST textures and LINEAR textures sampled at
 *     texel centres are integer stencils with index clamping;
 *   - GLSL pow(x, 6.0) and pow(x, 2.0) (glsl/pre_quality.fs:109-114) are evaluated
 *     as products (x2 = x*x, x4 = x2*x2, x6 = x4*x2): IEEE-like behaviour for
 *     negative x (GLSL leaves it undefined);
 *   - normalize(v) = v / sqrt(dot(v,v)), dot and cross in the glm association.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

#ifdef _OPENMP
#include <omp.h>
#endif
/* cpu_baseline threading: rows / z-slices are independent, so the loops below
 * carry `omp parallel for`; results do not depend on the thread count. */
ORC_API int orc_set_threads(int n)
{
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
  return omp_get_max_threads();
#else
  (void)n;
  return 1;
#endif
}

/* ------------------------------------------------------------------------- */
/* sampling (GL 4.4 section 8.14; sampler state table in SURVEY.md section 8a) */

static inline float lerpf(float a, float b, float t) { return a + t * (b - a); }

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* float floor value -> int index without undefined behaviour for NaN / huge */
static inline int idx_from_floor(float f, int n)
{
  float fc = fminf(fmaxf(f, -1.0f), (float)n); /* NaN -> -1 */
  return (int)fc;
}

/* Driver-tolerance study (never set by a parity test): GL 4.4 section 8.14 lets an
 * implementation compute the LINEAR weights in fixed point, and NVIDIA's texture units do -- the fraction of the texel
 * coordinate is held in 9-bit fixed point with 8 fractional bits (1.0 representable; CUDA C Programming Guide,
 * "Linear Filtering").  With g_weight_bits > 0 every LINEAR weight of the path (cv_xyz / cv_uv / cv_xyz_inv lookups,
 * framework/calibration/CalibVolumes.cpp:76,135,140; the colour, quality and silhouette fetches, NetKinectArray.cpp:46-53)
 * is rounded to that many fractional bits, so that exact-weight and fixed-point-weight runs of the same frame can be held
 * against each other: the bound a maintainer comparing against the authors' driver should expect. */
static int g_weight_bits = 0;
ORC_API void orc_set_linear_weight_bits(int bits) { g_weight_bits = bits < 0 ? 0 : (bits > 23 ? 23 : bits); }
ORC_API int orc_get_linear_weight_bits(void) { return g_weight_bits; }

static inline void axis_linear(float s, int n, int* i0, int* i1, float* a)
{
  float t = s * (float)n - 0.5f;
  float f = floorf(t);
  int j = idx_from_floor(f, n);
  *a = t - f;
  if (g_weight_bits > 0) {
    const float scale = (float)(1 << g_weight_bits);
    *a = rintf(*a * scale) / scale; /* NaN stays NaN */
  }
  *i0 = clampi(j, 0, n - 1);
  *i1 = clampi(j + 1, 0, n - 1);
}

static inline int axis_nearest(float s, int n)
{
  float f = floorf(s * (float)n);
  return clampi(idx_from_floor(f, n), 0, n - 1);
}

/* 3-D LINEAR / CLAMP_TO_EDGE lookup of a `ch`-channel record volume, x fastest
 * (framework/calibration/calibration_volume.hpp:57-59; texture state
 * framework/calibration/CalibVolumes.cpp:76,135,140) */
static void tex3d_linear(const float* vol, int ch, int nch, int rx, int ry, int rz,
                         float u, float v, float w, float* out)
{
  int x0, x1, y0, y1, z0, z1;
  float ax, ay, az;
  axis_linear(u, rx, &x0, &x1, &ax);
  axis_linear(v, ry, &y0, &y1, &ay);
  axis_linear(w, rz, &z0, &z1, &az);
  const size_t sx = (size_t)ch, sy = (size_t)rx * ch, sz = (size_t)rx * ry * ch;
  for (int c = 0; c < nch; ++c) {
    float t000 = vol[z0 * sz + y0 * sy + x0 * sx + c];
    float t100 = vol[z0 * sz + y0 * sy + x1 * sx + c];
    float t010 = vol[z0 * sz + y1 * sy + x0 * sx + c];
    float t110 = vol[z0 * sz + y1 * sy + x1 * sx + c];
    float t001 = vol[z1 * sz + y0 * sy + x0 * sx + c];
    float t101 = vol[z1 * sz + y0 * sy + x1 * sx + c];
    float t011 = vol[z1 * sz + y1 * sy + x0 * sx + c];
    float t111 = vol[z1 * sz + y1 * sy + x1 * sx + c];
    float c00 = lerpf(t000, t100, ax);
    float c10 = lerpf(t010, t110, ax);
    float c01 = lerpf(t001, t101, ax);
    float c11 = lerpf(t011, t111, ax);
    float c0 = lerpf(c00, c10, ay);
    float c1 = lerpf(c01, c11, ay);
    out[c] = lerpf(c0, c1, az);
  }
}

/* 2-D LINEAR lookup, `ch` interleaved float channels */
static void tex2d_linear(const float* img, int ch, int nch, int W, int H, float u, float v, float* out)
{
  int x0, x1, y0, y1;
  float ax, ay;
  axis_linear(u, W, &x0, &x1, &ax);
  axis_linear(v, H, &y0, &y1, &ay);
  for (int c = 0; c < nch; ++c) {
    float t00 = img[((size_t)y0 * W + x0) * ch + c];
    float t10 = img[((size_t)y0 * W + x1) * ch + c];
    float t01 = img[((size_t)y1 * W + x0) * ch + c];
    float t11 = img[((size_t)y1 * W + x1) * ch + c];
    out[c] = lerpf(lerpf(t00, t10, ax), lerpf(t01, t11, ax), ay);
  }
}

/* colour array: GL_RGB / GL_UNSIGNED_BYTE -> normalised [0,1], LINEAR
 * (framework/NetKinectArray.cpp:157-160, framework/rendering/TextureArray.cpp:27) */
static void tex2d_linear_rgb8(const uint8_t* img, int W, int H, float u, float v, float* out)
{
  int x0, x1, y0, y1;
  float ax, ay;
  axis_linear(u, W, &x0, &x1, &ax);
  axis_linear(v, H, &y0, &y1, &ay);
  for (int c = 0; c < 3; ++c) {
    float t00 = (float)img[((size_t)y0 * W + x0) * 3 + c] / 255.0f;
    float t10 = (float)img[((size_t)y0 * W + x1) * 3 + c] / 255.0f;
    float t01 = (float)img[((size_t)y1 * W + x0) * 3 + c] / 255.0f;
    float t11 = (float)img[((size_t)y1 * W + x1) * 3 + c] / 255.0f;
    out[c] = lerpf(lerpf(t00, t10, ax), lerpf(t01, t11, ax), ay);
  }
}

ORC_API void orc_tex3d_linear(const float* vol, int ch, int rx, int ry, int rz,
                              float u, float v, float w, float* out)
{
  tex3d_linear(vol, ch, ch, rx, ry, rz, u, v, w, out);
}

ORC_API void orc_tex2d_linear(const float* img, int ch, int W, int H, float u, float v, float* out)
{
  tex2d_linear(img, ch, ch, W, H, u, v, out);
}

ORC_API void orc_tex2d_linear_rgb8(const uint8_t* img, int W, int H, float u, float v, float* out)
{
  tex2d_linear_rgb8(img, W, H, u, v, out);
}

ORC_API int orc_axis_nearest(float s, int n) { return axis_nearest(s, n); }

/* ------------------------------------------------------------------------- */
/* small vector helpers in the glm association (external/glm-0.9.5.3)         */

static inline float dot3(const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

static inline void cross3(const float* x, const float* y, float* o)
{
  o[0] = x[1] * y[2] - y[1] * x[2];
  o[1] = x[2] * y[0] - y[2] * x[0];
  o[2] = x[0] * y[1] - y[0] * x[1];
}

static inline void normalize3(const float* v, float* o)
{
  float l = sqrtf(dot3(v, v));
  o[0] = v[0] / l;
  o[1] = v[1] / l;
  o[2] = v[2] / l;
}

static inline float distance3(const float* a, const float* b)
{
  float d[3] = {a[0] - b[0], a[1] - b[1], a[2] - b[2]};
  return sqrtf(dot3(d, d));
}

/* GLSL pow with the constant exponents the path uses (header note) */
ORC_API float orc_pow6(float x)
{
  float x2 = x * x;
  float x4 = x2 * x2;
  return x4 * x2;
}
ORC_API float orc_pow2(float x) { return x * x; }

/* ------------------------------------------------------------------------- */
/* pre_morph.fs (a2): pass mode 0 = dilate(), mode 1 = copy                   */
/* glsl/pre_morph.fs:73-112 (dilate), :114-140 (main); host                   */
/* framework/NetKinectArray.cpp:251-290.  in_bbox(texcoord, depth) always     */
/* returns true (:44-50), so the cv_xyz fetch there is dead.                  */

static inline int morph_valid(float d) { return d > 0.5f && d < 4.5f; } /* :36-39 */

ORC_API void orc_morph(const float* in, int W, int H, unsigned mode, float* out)
{
#pragma omp parallel for schedule(static)
  for (int py = 0; py < H; ++py) {
    for (int px = 0; px < W; ++px) {
      float depth = in[(size_t)py * W + px];
      float res;
      if (mode == 1u) {
        res = depth;
      } else if (mode != 0u) {
        res = 0.25f;
      } else if (morph_valid(depth)) {
        res = depth;
      } else {
        float average_depth = 0.0f;
        int valid = 0;
        float num_samples = 0.0f;
        for (int y = -1; y < 2; ++y) {
          for (int x = -1; x < 2; ++x) {
            float ds = in[(size_t)clampi(py + y, 0, H - 1) * W + clampi(px + x, 0, W - 1)];
            if (morph_valid(ds)) {
              valid = 1;
              average_depth += ds;
              num_samples += 1.0f;
            }
          }
        }
        if (!valid) {
          res = 0.0f;
        } else {
          average_depth /= num_samples;
          float new_depth = 0.0f;
          num_samples = 0.0f;
          valid = 0;
          for (int y = -1; y < 2; ++y) {
            for (int x = -1; x < 2; ++x) {
              float ds = in[(size_t)clampi(py + y, 0, H - 1) * W + clampi(px + x, 0, W - 1)];
              if (morph_valid(ds) && fabsf(average_depth - ds) < 0.2f) {
                valid = 1;
                new_depth += ds;
                num_samples += 1.0f;
              }
            }
          }
          res = valid ? new_depth / num_samples : 0.0f;
        }
      }
      out[(size_t)py * W + px] = res;
    }
  }
}

/* ------------------------------------------------------------------------- */
/* inc_color.glsl (Lab conversion incl. the extra /255, :14-16)               */

static inline float pivot_rgb(float n)
{
  return (n > 0.04045f ? powf((n + 0.055f) / 1.055f, 2.4f) : n / 12.92f) * 100.0f;
}
static inline float pivot_xyz(float n)
{
  return n > 0.008856f ? powf(n, 1.0f / 3.0f) : (903.3f * n + 16.0f) / 116.0f;
}
ORC_API void orc_rgb_to_lab(const float* rgb, float* lab)
{
  float r = pivot_rgb(rgb[0] / 255.0f);
  float g = pivot_rgb(rgb[1] / 255.0f);
  float b = pivot_rgb(rgb[2] / 255.0f);
  float X = r * 0.4124f + g * 0.3576f + b * 0.1805f;
  float Y = r * 0.2126f + g * 0.7152f + b * 0.0722f;
  float Z = r * 0.0193f + g * 0.1192f + b * 0.9505f;
  float x = pivot_xyz(X / 95.047f);
  float y = pivot_xyz(Y / 100.000f);
  float z = pivot_xyz(Z / 108.883f);
  lab[0] = fmaxf(0.0f, 116.0f * y - 16.0f);
  lab[1] = 500.0f * (x - y);
  lab[2] = 200.0f * (y - z);
}

/* ------------------------------------------------------------------------- */
/* DXT1 / DXT5 colour frames -> RGB8.  The GL driver decodes the compressed    */
/* colour array in the reference (NetKinectArray.cpp:149-156); the reference's  */
/* own CPU decode of the same frames is squish::DecompressImage                 */
/* (NetKinectArray.cpp:633, external/squish/colourblock.cpp:140-214), whose     */
/* integer arithmetic is restated here and pinned against it in oracle/_ref.    */
/* mode 1 = DXT1 (8-B blocks), mode 5 = DXT5 (16-B blocks, colour at +8).      */

static void unpack565(const uint8_t* p, uint8_t* c)
{
  int value = (int)p[0] | ((int)p[1] << 8);
  uint8_t r = (uint8_t)((value >> 11) & 0x1f), g = (uint8_t)((value >> 5) & 0x3f), b = (uint8_t)(value & 0x1f);
  c[0] = (uint8_t)((r << 3) | (r >> 2));
  c[1] = (uint8_t)((g << 2) | (g >> 4));
  c[2] = (uint8_t)((b << 3) | (b >> 2));
}

ORC_API void orc_decode_dxt(const uint8_t* blocks, int W, int H, int mode, uint8_t* rgb)
{
  const int bpb = mode == 1 ? 8 : 16;
  const uint8_t* src = blocks;
  for (int y = 0; y < H; y += 4) {
    for (int x = 0; x < W; x += 4) {
      const uint8_t* cb = src + (mode == 1 ? 0 : 8);
      uint8_t codes[4][3];
      unpack565(cb, codes[0]);
      unpack565(cb + 2, codes[1]);
      int a = (int)cb[0] | ((int)cb[1] << 8), b = (int)cb[2] | ((int)cb[3] << 8);
      for (int i = 0; i < 3; ++i) {
        int c = codes[0][i], d = codes[1][i];
        if (mode == 1 && a <= b) {
          codes[2][i] = (uint8_t)((c + d) / 2);
          codes[3][i] = 0;
        } else {
          codes[2][i] = (uint8_t)((2 * c + d) / 3);
          codes[3][i] = (uint8_t)((c + 2 * d) / 3);
        }
      }
      for (int py = 0; py < 4; ++py)
        for (int px = 0; px < 4; ++px) {
          int sx = x + px, sy = y + py;
          if (sx >= W || sy >= H) continue;
          int idx = (cb[4 + py] >> (2 * px)) & 3;
          for (int k = 0; k < 3; ++k) rgb[((size_t)sy * W + sx) * 3 + k] = codes[idx][k];
        }
      src += bpb;
    }
  }
}

/* ------------------------------------------------------------------------- */
/* pre_depth.fs (a3): bilateral filter + Lab colour                           */

typedef struct {
  int W, H;             /* depth resolution */
  int Wc, Hc;           /* colour resolution */
  int xyz_res[3];       /* cv_xyz / cv_uv resolution (same header, CalibVolumes.cpp:115-130) */
  int uv_res[3];
  float cv_min_ds, cv_max_ds; /* depthLimits() of cv_xyz (NetKinectArray.cpp:339-340) */
  float bbox_min[3], bbox_max[3];
  int filter_textures;  /* m_filter_textures */
  int compress;         /* compress_depth: input is the u8 texture normalised to [0,1] */
  float near_, far_;    /* yml near_far (NetKinectArray.cpp:346-351) */
} orc_depth_params;

static inline float pd_sample(const float* depth, int W, int H, int px, int py, const orc_depth_params* p)
{
  float d = depth[(size_t)clampi(py, 0, H - 1) * W + clampi(px, 0, W - 1)];
  if (p->compress) { /* pre_depth.fs:51-61 */
    float scale = p->far_ - p->near_;
    float scaled_near = scale / 255.0f;
    if (d < scaled_near) return 0.0f;
    return (d * d + 0.15f * scaled_near) * scale + p->near_;
  }
  return d;
}

ORC_API void orc_pre_depth(const float* depth, const uint8_t* color_rgb8, const float* cv_xyz,
                           const float* cv_uv, const orc_depth_params* p, float* out_depth_rg,
                           float* out_lab)
{
  const int W = p->W, H = p->H;
  const float inv_k = 1.0f / 6.0f; /* dist_space_max_inv, pre_depth.fs:37 */
#pragma omp parallel for schedule(dynamic, 4)
  for (int py = 0; py < H; ++py) {
    for (int px = 0; px < W; ++px) {
      const size_t o = (size_t)py * W + px;
      const float u = ((float)px + 0.5f) / (float)W;
      const float v = ((float)py + 0.5f) / (float)H;
      float depth0 = pd_sample(depth, W, H, px, py, p);
      float range = p->cv_max_ds - p->cv_min_ds;
      float depth_norm = (depth0 - p->cv_min_ds) / range;
      float pos_world[3];
      tex3d_linear(cv_xyz, 3, 3, p->xyz_res[0], p->xyz_res[1], p->xyz_res[2], u, v, depth_norm, pos_world);
      int in_box = pos_world[0] >= p->bbox_min[0] && pos_world[1] >= p->bbox_min[1] &&
                   pos_world[2] >= p->bbox_min[2] && pos_world[0] <= p->bbox_max[0] &&
                   pos_world[1] <= p->bbox_max[1] && pos_world[2] <= p->bbox_max[2];
      /* pre_depth.fs:136 */
      float dn_c = (depth_norm <= 0.0f || depth_norm >= 1.0f) ? 1.0f : depth_norm;
      float cc[2], rgb[3];
      tex3d_linear(cv_uv, 2, 2, p->uv_res[0], p->uv_res[1], p->uv_res[2], u, v, dn_c, cc);
      tex2d_linear_rgb8(color_rgb8, p->Wc, p->Hc, cc[0], cc[1], rgb);
      orc_rgb_to_lab(rgb, out_lab + o * 3);
      if (!in_box) {
        out_depth_rg[o * 2 + 0] = 0.0f;
        out_depth_rg[o * 2 + 1] = 0.0f;
        continue;
      }
      if (!p->filter_textures) {
        out_depth_rg[o * 2 + 0] = depth_norm;
        out_depth_rg[o * 2 + 1] = 1.0f;
        continue;
      }
      /* bilateral_filter, pre_depth.fs:85-127 */
      float d_dmax = depth0 / 4.5f;
      float dist_range_max = 0.35f * d_dmax;
      float dist_range_max_inv = 1.0f / dist_range_max;
      float depth_bf = 0.0f, w = 0.0f, w_range = 0.0f, num_samples = 0.0f;
      for (int y = -6; y < 7; ++y) {
        for (int x = -6; x < 7; ++x) {
          num_samples += 1.0f;
          float depth_s = pd_sample(depth, W, H, px + x, py + y, p);
          float depth_range = fabsf(depth_s - depth0);
          if ((depth_s < p->cv_min_ds) || (depth_s > p->cv_max_ds) || (depth_range > dist_range_max)) continue;
          float len = sqrtf((float)x * (float)x + (float)y * (float)y);
          float gauss_space = 1.0f - len * inv_k;
          float gauss_range = 1.0f - fminf(depth_range, dist_range_max) * dist_range_max_inv;
          float w_s = gauss_space * gauss_range;
          depth_bf += w_s * depth_s;
          w += w_s;
          w_range += gauss_range;
        }
      }
      float filtered = depth_bf / w;
      out_depth_rg[o * 2 + 0] = (filtered - p->cv_min_ds) / range;
      out_depth_rg[o * 2 + 1] = w_range / num_samples;
    }
  }
}

/* ------------------------------------------------------------------------- */
/* pre_boundary.fs (a4)                                                        */

ORC_API void orc_boundary(const float* depth_rg, const float* lab, int W, int H, int refine,
                          float* out_depth_b_rg, float* out_sil)
{
#pragma omp parallel for schedule(dynamic, 4)
  for (int py = 0; py < H; ++py) {
    for (int px = 0; px < W; ++px) {
      const size_t o = (size_t)py * W + px;
      float dx = depth_rg[o * 2 + 0], dy = depth_rg[o * 2 + 1];
      float sil = 1.0f;
      if (dx <= 0.0f) { /* :90-100 */
        dy = 0.0f;
        sil = 0.0f;
      } else if (!(dy > 0.65f)) { /* :102-113 */
        sil = 0.0f;
        /* get_color_diff, :37-55 */
        const float* color = lab + o * 3;
        float total_dist = 0.0f, num_samples = 0.0f;
        for (int y = -2; y < 3; ++y) {
          for (int x = -2; x < 3; ++x) {
            size_t os = (size_t)clampi(py + y, 0, H - 1) * W + clampi(px + x, 0, W - 1);
            if (depth_rg[os * 2 + 0] > 0.0f && depth_rg[os * 2 + 1] > 0.65f) {
              num_samples += 1.0f;
              total_dist += distance3(color, lab + os * 3);
            }
          }
        }
        float color_dist = (num_samples < 16.0f * 0.5f) ? 1.0f : total_dist / num_samples;
        if (color_dist > 0.5f || !refine) {
          dx = -1.0f;
          dy = 0.1f;
        } else {
          dy = 1.0f;
        }
      } else {
        dy = 0.0f;
      }
      out_depth_b_rg[o * 2 + 0] = dx;
      out_depth_b_rg[o * 2 + 1] = dy;
      out_sil[o] = sil;
    }
  }
}

/* ------------------------------------------------------------------------- */
/* inc_bricks.glsl mark_brick (:40-58) + pre_normal.fs (a5)                   */

typedef struct {
  int W, H;
  int xyz_res[3];
  float bbox_min[3], bbox_max[3];
  float brick_size;    /* metres, ReconIntegration::setBrickSize */
  int res_bricks[3];   /* m_res_bricks from divideBox */
} orc_normal_params;

static inline float signf_(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }

/* Returns 0 if the home brick index is outside the brick grid: the reference
 * converts a possibly negative float to uvec3 there (undefined in GLSL) and
 * indexes out of range; this build SKIPS such positions. */
static int mark_brick(const float* pos, const orc_normal_params* p, uint32_t* bricks)
{
  int idx[3];
  float diff[3], dabs[3];
  for (int a = 0; a < 3; ++a) {
    float f = floorf((pos[a] - p->bbox_min[a]) / p->brick_size);
    if (!(f >= 0.0f) || !(f < (float)p->res_bricks[a])) return 0;
    idx[a] = (int)f;
  }
  for (int a = 0; a < 3; ++a) {
    /* to_world(vec3(0.5), index) = vec3(index)*brick_size + bbox_min + 0.5*brick_size */
    float center = (float)idx[a] * p->brick_size + p->bbox_min[a] + 0.5f * p->brick_size;
    diff[a] = pos[a] - center;
    dabs[a] = fabsf(diff[a]);
  }
  float min_v = fmaxf(dabs[0], fmaxf(dabs[1], dabs[2]));
  int nb[3];
  for (int a = 0; a < 3; ++a) {
    float mc = (dabs[a] < min_v) ? 0.0f : 1.0f;
    int off = (int)signf_(diff[a] * mc);
    nb[a] = clampi(idx[a] + off, 0, p->res_bricks[a] - 1);
  }
  const int rx = p->res_bricks[0], ry = p->res_bricks[1];
  const uint32_t inc = (dabs[0] > p->brick_size * 0.1f) ? 1u : 0u;
#pragma omp atomic
  bricks[(size_t)nb[2] * ry * rx + (size_t)nb[1] * rx + nb[0]] += inc;
#pragma omp atomic
  bricks[(size_t)idx[2] * ry * rx + (size_t)idx[1] * rx + idx[0]] += 1u;
  return 1;
}

static inline int unit_outside(float d) { return (d <= 0.0f) || (d >= 1.0f); }

ORC_API void orc_normal(const float* depth_b_rg, const float* cv_xyz, const orc_normal_params* p,
                        float* out_normal, uint32_t* bricks /* may be NULL */)
{
  const int W = p->W, H = p->H;
  const float tsx = 1.0f / (float)W, tsy = 1.0f / (float)H; /* texSizeInv NetKinectArray.cpp:197 */
#pragma omp parallel for schedule(dynamic, 4)
  for (int py = 0; py < H; ++py) {
    for (int px = 0; px < W; ++px) {
      const size_t o = (size_t)py * W + px;
      float* n = out_normal + o * 3;
      float depth = depth_b_rg[o * 2];
      if (unit_outside(depth)) {
        n[0] = n[1] = n[2] = 0.0f;
        continue;
      }
      const float u = ((float)px + 0.5f) / (float)W;
      const float v = ((float)py + 0.5f) / (float)H;
      float world[3];
      tex3d_linear(cv_xyz, 3, 3, p->xyz_res[0], p->xyz_res[1], p->xyz_res[2], u, v, depth, world);
      if (bricks) mark_brick(world, p, bricks);
      /* pre_normal.fs:35-55; "t" is +y, "b" is -y */
      float vt = v + tsy, vb = v - tsy, ul = u - tsx, ur = u + tsx;
      float dt = depth_b_rg[((size_t)clampi(py + 1, 0, H - 1) * W + px) * 2];
      float db = depth_b_rg[((size_t)clampi(py - 1, 0, H - 1) * W + px) * 2];
      float dl = depth_b_rg[((size_t)py * W + clampi(px - 1, 0, W - 1)) * 2];
      float dr = depth_b_rg[((size_t)py * W + clampi(px + 1, 0, W - 1)) * 2];
      dt = unit_outside(dt) ? depth : dt;
      db = unit_outside(db) ? depth : db;
      dl = unit_outside(dl) ? depth : dl;
      dr = unit_outside(dr) ? depth : dr;
      float wt[3], wb[3], wl[3], wr[3];
      tex3d_linear(cv_xyz, 3, 3, p->xyz_res[0], p->xyz_res[1], p->xyz_res[2], u, vt, dt, wt);
      tex3d_linear(cv_xyz, 3, 3, p->xyz_res[0], p->xyz_res[1], p->xyz_res[2], u, vb, db, wb);
      tex3d_linear(cv_xyz, 3, 3, p->xyz_res[0], p->xyz_res[1], p->xyz_res[2], ul, v, dl, wl);
      tex3d_linear(cv_xyz, 3, 3, p->xyz_res[0], p->xyz_res[1], p->xyz_res[2], ur, v, dr, wr);
      float a[3] = {wb[0] - wt[0], wb[1] - wt[1], wb[2] - wt[2]};
      float b[3] = {wl[0] - wr[0], wl[1] - wr[1], wl[2] - wr[2]};
      float c[3];
      cross3(a, b, c);
      normalize3(c, n);
    }
  }
}

/* ------------------------------------------------------------------------- */
/* pre_quality.fs (a6)                                                         */

ORC_API void orc_quality(const float* depth_b_rg, const float* normals, const float* cv_xyz,
                         const int* xyz_res, const float* cam_pos, int W, int H, float* out_quality)
{
  const float inv_k = 1.0f / 6.0f;
  (void)inv_k; /* gauss_space feeds only depth_bf / w, which the shader discards */
#pragma omp parallel for schedule(dynamic, 4)
  for (int py = 0; py < H; ++py) {
    for (int px = 0; px < W; ++px) {
      const size_t o = (size_t)py * W + px;
      float depth = depth_b_rg[o * 2];
      if (unit_outside(depth)) {
        out_quality[o] = 0.0f;
        continue;
      }
      float dist_range_max = 0.35f * (depth / 1.0f);
      float dist_range_max_inv = 1.0f / dist_range_max;
      float w_range = 0.0f, border_samples = 0.0f, num_samples = 0.0f;
      for (int y = -6; y < 7; ++y) {
        for (int x = -6; x < 7; ++x) {
          num_samples += 1.0f;
          float ds = depth_b_rg[((size_t)clampi(py + y, 0, H - 1) * W + clampi(px + x, 0, W - 1)) * 2];
          float depth_range = fabsf(ds - depth);
          if (unit_outside(ds) || (depth_range > dist_range_max)) {
            border_samples += 1.0f;
            continue;
          }
          float gauss_range = 1.0f - fminf(depth_range, dist_range_max) * dist_range_max_inv;
          w_range += gauss_range;
        }
      }
      float lateral_quality = 1.0f - border_samples / num_samples;
      float q = orc_pow6(lateral_quality);
      q *= orc_pow6(w_range / num_samples);
      q /= depth * 6.5f;
      /* normal_angle, :43-48 */
      const float u = ((float)px + 0.5f) / (float)W;
      const float v = ((float)py + 0.5f) / (float)H;
      float wp[3], d[3], dn[3];
      tex3d_linear(cv_xyz, 3, 3, xyz_res[0], xyz_res[1], xyz_res[2], u, v, depth, wp);
      d[0] = cam_pos[0] - wp[0];
      d[1] = cam_pos[1] - wp[1];
      d[2] = cam_pos[2] - wp[2];
      normalize3(d, dn);
      float angle = dot3(dn, normals + o * 3);
      q *= orc_pow2(angle);
      out_quality[o] = q;
    }
  }
}

/* ------------------------------------------------------------------------- */
/* tsdf_integration.vs (a9) + VolumeSampler voxel positions (a8)              */
/* Writes voxels z in [z0, z1) of an X*Y*Z x-fastest volume (whole-volume     */
/* pointer).  `voxel_mask`, if not NULL, is one byte per voxel of rows         */
/* [z0, z1) (orc_brick_voxel_mask below: the voxels in the index list of an    */
/* occupied brick); the others keep the clear value -limit                     */
/* (framework/reconstruction/recon_integration.cpp:243-270).                  */

typedef struct {
  int num_sensors;
  int W, H;
  int res[3];          /* TSDF resolution m_res_volume */
  float limit;
} orc_integrate_params;

/* The voxels the reference draws in brick mode (recon_integration.cpp:255-259:
 * for every occupied brick, m_sampler.sample(m_bricks[index].indices)): the three
 * nested while loops of divideBox (:366-388, brick id = emplace order, x fastest)
 * and, per brick, the index list of VolumeSampler::containedVoxels
 * (volume_sampler.cpp:50-62) -- binary32 pos / step truncated to unsigned, float
 * upper bounds, linear index z*X*Y + y*X + x.  An index triple past the x or y end
 * of the volume therefore aliases a voxel of the next row / slice; an index beyond
 * X*Y*Z leaves the vertex buffer (undefined in GL) and is dropped here.
 * out: one byte per voxel of rows [z0, z1), set to 1 where some occupied brick
 * lists the voxel.  res_bricks receives m_res_bricks.  Returns the number of listed
 * index triples that lie outside the volume on some axis (aliased or dropped). */
ORC_API size_t orc_brick_voxel_mask(const float* bbox_min, const float* bbox_max, float brick_size, const int* dims,
                                    const uint8_t* occupied_brick_mask, int z0, int z1, uint8_t* out, int* res_bricks)
{
  const float size[3] = {bbox_max[0] - bbox_min[0], bbox_max[1] - bbox_min[1], bbox_max[2] - bbox_min[2]};
  const float stepv[3] = {1.0f / (float)dims[0], 1.0f / (float)dims[1], 1.0f / (float)dims[2]};
  const size_t XY = (size_t)dims[0] * dims[1], total = XY * (size_t)dims[2];
  memset(out, 0, XY * (size_t)(z1 - z0));
  size_t brick = 0, outside = 0;
  int rb[3] = {0, 0, 0};
  float start[3] = {bbox_min[0], bbox_min[1], bbox_min[2]};
  while (size[2] - start[2] + bbox_min[2] > 0.0f) {
    while (size[1] - start[1] + bbox_min[1] > 0.0f) {
      while (size[0] - start[0] + bbox_min[0] > 0.0f) {
        float pos[3], bs[3];
        for (int a = 0; a < 3; ++a) {
          bs[a] = fminf(brick_size, size[a] - start[a] + bbox_min[a]) / size[a];
          pos[a] = (start[a] - bbox_min[a]) / size[a];
        }
        if (occupied_brick_mask[brick]) {
          for (unsigned y = (unsigned)(pos[1] / stepv[1]); (float)y < (pos[1] + bs[1]) / stepv[1]; ++y)
            for (unsigned x = (unsigned)(pos[0] / stepv[0]); (float)x < (pos[0] + bs[0]) / stepv[0]; ++x)
              for (unsigned z = (unsigned)(pos[2] / stepv[2]); (float)z < (pos[2] + bs[2]) / stepv[2]; ++z) {
                const size_t id = (size_t)z * XY + (size_t)y * (unsigned)dims[0] + x;
                if (x >= (unsigned)dims[0] || y >= (unsigned)dims[1] || z >= (unsigned)dims[2]) ++outside;
                if (id >= total) continue;
                if (id >= (size_t)z0 * XY && id < (size_t)z1 * XY) out[id - (size_t)z0 * XY] = 1;
              }
        }
        ++brick;
        start[0] += brick_size;
        if (rb[2] == 0 && rb[1] == 0) ++rb[0];
      }
      start[0] = bbox_min[0];
      start[1] += brick_size;
      if (rb[2] == 0) ++rb[1];
    }
    start[1] = bbox_min[1];
    start[2] += brick_size;
    ++rb[2];
  }
  if (res_bricks) {
    res_bricks[0] = rb[0];
    res_bricks[1] = rb[1];
    res_bricks[2] = rb[2];
  }
  return outside;
}

ORC_API void orc_integrate(const orc_integrate_params* p, const float* const* cv_xyz_inv /* RGBA */,
                           const int* inv_res /* 3 per sensor */, const float* const* silhouette,
                           const float* const* depth_b_rg, const float* const* quality,
                           const uint8_t* voxel_mask, int z0, int z1, float* tsdf)
{
  const int X = p->res[0], Y = p->res[1], Z = p->res[2];
  const float stepX = 1.0f / (float)X, stepY = 1.0f / (float)Y, stepZ = 1.0f / (float)Z;
  const float limit = p->limit;
  (void)Z;
#pragma omp parallel for collapse(2) schedule(dynamic, 4)
  for (int z = z0; z < z1; ++z) {
    for (int y = 0; y < Y; ++y) {
      for (int x = 0; x < X; ++x) {
        const size_t o = (size_t)z * X * Y + (size_t)y * X + x;
        if (voxel_mask && !voxel_mask[o - (size_t)z0 * X * Y]) { /* not drawn: keeps the clear value */
          tsdf[o] = -limit;
          continue;
        }
        /* framework/rendering/volume_sampler.cpp:36-42 */
        const float pos[3] = {((float)x + 0.5f) * stepX, ((float)y + 0.5f) * stepY, ((float)z + 0.5f) * stepZ};
        float weighted_tsd = limit;
        float total_weight = 0.0f;
        for (int i = 0; i < p->num_sensors; ++i) {
          float pc[3];
          tex3d_linear(cv_xyz_inv[i], 4, 3, inv_res[3 * i], inv_res[3 * i + 1], inv_res[3 * i + 2],
                       pos[0], pos[1], pos[2], pc);
          float sil;
          tex2d_linear(silhouette[i], 1, 1, p->W, p->H, pc[0], pc[1], &sil);
          if (sil < 1.0f) {
            if (weighted_tsd >= limit) {
              weighted_tsd = -limit;
              continue;
            }
          }
          int ix = axis_nearest(pc[0], p->W), iy = axis_nearest(pc[1], p->H);
          float depth = depth_b_rg[i][((size_t)iy * p->W + ix) * 2];
          float sdist = pc[2] - depth;
          if (sdist <= -limit) {
            weighted_tsd = -limit;
          } else if (sdist >= limit) {
          } else {
            float weight;
            tex2d_linear(quality[i], 1, 1, p->W, p->H, pc[0], pc[1], &weight);
            weighted_tsd = (weighted_tsd * total_weight + weight * sdist) / (total_weight + weight);
            total_weight += weight;
          }
        }
        tsdf[o] = weighted_tsd;
      }
    }
  }
}

/* ------------------------------------------------------------------------- */
/* Frustum::getCameraPos (framework/calibration/frustum.cpp:21-33, :97-111)   */
/* corners from getCornerPoints (framework/calibration/CalibVolumes.cpp:98-113) */

static void closest_point(const float* p, const float* u, const float* q, const float* v, float* o)
{
  float w0[3] = {p[0] - q[0], p[1] - q[1], p[2] - q[2]};
  float a = dot3(u, u), b = dot3(u, v), c = dot3(v, v), d = dot3(u, w0), e = dot3(v, w0);
  float sc = (b * e - c * d) / (a * c - b * b);
  float tc = (a * e - b * d) / (a * c - b * b);
  for (int k = 0; k < 3; ++k) {
    float pc = p[k] + u[k] * sc;
    float qc = q[k] + v[k] * tc;
    o[k] = (pc + qc) * 0.5f;
  }
}

ORC_API void orc_camera_pos(const float* cv_xyz, const int* res, float* out)
{
  const int ex = res[0] - 1, ey = res[1] - 1, ez = res[2] - 1;
  const int cx[8] = {0, 0, ex, ex, 0, 0, ex, ex};
  const int cy[8] = {0, ey, ey, 0, 0, ey, ey, 0};
  const int cz[8] = {0, 0, 0, 0, ez, ez, ez, ez};
  float c[8][3];
  for (int i = 0; i < 8; ++i)
    for (int k = 0; k < 3; ++k)
      c[i][k] = cv_xyz[(((size_t)cz[i] * res[1] + cy[i]) * res[0] + cx[i]) * 3 + k];
  float cn[3], cf[3], vd[3];
  for (int k = 0; k < 3; ++k) {
    cn[k] = (c[0][k] + c[1][k] + c[2][k] + c[3][k]) / 4.0f;
    cf[k] = (c[4][k] + c[5][k] + c[6][k] + c[7][k]) / 4.0f;
    vd[k] = cf[k] - cn[k];
  }
  float pts[4][3];
  for (int i = 0; i < 4; ++i) {
    float u[3] = {c[i][0] - c[i + 4][0], c[i][1] - c[i + 4][1], c[i][2] - c[i + 4][2]};
    closest_point(c[i], u, cn, vd, pts[i]);
  }
  for (int k = 0; k < 3; ++k) out[k] = (pts[0][k] + pts[1][k] + pts[2][k] + pts[3][k]) / 4.0f;
}

/* ------------------------------------------------------------------------- */
/* tsdf_raymarch.{vs,fs} + shading.glsl (f-2): one ray per pixel of the viewport. */
/* The reference rasterises the unit cube and interpolates pass_Position          */
/* (tsdf_raymarch.vs:13-16); every fragment of a pixel marches the same ray from  */
/* CameraPos, so this restatement builds that ray from the pixel centre through    */
/* the far-plane point screenToVol((px+.5, py+.5, 1)) (:384-390) and treats a pixel */
/* as covered when the ray meets the unit cube in front of the camera.  All        */
/* matrices are the host-side uniforms ReconIntegration::draw uploads              */
/* (recon_integration.cpp:177-241), column-major, passed in verbatim.              */

typedef struct {
  float modelview[16], projection[16];
  float normal_matrix[16];         /* inverseTranspose(modelview * vol_to_world) */
  float gl_normal_matrix_inv[16];  /* inverse(gl_NormalMatrix), shade mode 2 */
  float vol_to_world[16], vol_to_world_inv[16], modelview_inv[16];
  float img_to_eye[16];            /* inverse(viewport_scale * viewport_translate * projection) */
  float camera_pos[3];             /* volume space */
  int width, height;
  int shade_mode;
  int skip_space;                  /* start positions from the brick depth peels (getStartPos) */
} orc_view;

typedef struct {
  int num_sensors, W, H, Wc, Hc;
  int res[3];                      /* TSDF */
  float limit;
} orc_raymarch_params;

static inline void mat4_mul_vec4(const float* m, const float* v, float* o)
{
  /* glm: m[0]*v.x + m[1]*v.y + (m[2]*v.z + m[3]*v.w), columns m[c] = m + 4c */
  for (int r = 0; r < 4; ++r) o[r] = (m[r] * v[0] + m[4 + r] * v[1]) + (m[8 + r] * v[2] + m[12 + r] * v[3]);
}

/* ---- brick depth peels (f-4): ReconIntegration::drawDepthLimits (recon_integration.cpp:409-429)
 * with glsl/bricks.{vs,gs,fs}: the occupied bricks are drawn as instanced unit cubes, no
 * culling, no depth test, MIN blending into an RGBA32F target cleared to (1,0,1,0); the
 * fragment shader writes (z, -z, front-facing ? 1 : z, 1) and the geometry shader drops a
 * face whose neighbour brick across it has a counter > 10 (inc_bricks.glsl:60-62).  Per pixel
 * that is: r = nearest emitted face, -g = farthest, b = nearest back face.  Restated per
 * ray: walk the brick grid along the pixel's ray; entering an occupied brick is a front
 * face, leaving one a back face (the unit-cube strip is wound counter-clockwise seen from
 * outside); faces outside the depth range [0,1] are clipped.  The neighbour across a face on
 * the grid's boundary: the shader adds ivec3(+-1) to the uvec3 brick index and linearises it
 * (bricks.gs:28-43, inc_bricks.glsl:25-27), so x = -1 wraps to the linear id before (the last
 * brick of the previous row), x = res.x to the id after, y = -1 / res.y to the ids res.x before /
 * after -- an ALIASED brick inside the buffer whose counter decides the cull; only ids before
 * the first or past the last brick are out of the buffer's range (robust access: 0 -> not
 * occupied).  Seen in the run of the shaders on Mesa (tests/test_gl_ref.py): whole boundary
 * faces are culled by the counter of the aliased brick. */

typedef struct {
  float bbox_min[3];
  float brick_size;
  int res_bricks[3];
} orc_brick_grid;

static inline float peel_z(const float* pmv, const float* o, const float* d, float t)
{
  const float w4[4] = {o[0] + d[0] * t, o[1] + d[1] * t, o[2] + d[2] * t, 1.0f};
  float c[4];
  mat4_mul_vec4(pmv, w4, c);
  return (c[2] / c[3]) * 0.5f + 0.5f;
}

static inline int grid_counter_gt10(const orc_brick_grid* g, const uint32_t* counters, const int* c)
{
  /* `c` may lie one cell outside the grid along one axis: the shader's uint arithmetic wraps to the same linear id */
  const long nb = (long)g->res_bricks[0] * g->res_bricks[1] * g->res_bricks[2];
  const long id = ((long)c[2] * g->res_bricks[1] + c[1]) * g->res_bricks[0] + c[0];
  if (id < 0 || id >= nb) return 0;
  return counters[id] > 10u;
}
static inline int grid_in_list(const orc_brick_grid* g, const uint8_t* mask, const int* c)
{
  if (c[0] < 0 || c[1] < 0 || c[2] < 0 || c[0] >= g->res_bricks[0] || c[1] >= g->res_bricks[1] || c[2] >= g->res_bricks[2]) return 0;
  return mask[((size_t)c[2] * g->res_bricks[1] + c[1]) * g->res_bricks[0] + c[0]] != 0;
}

/* one pixel: out = (r, g, b, a) */
static void depth_peel_pixel(const orc_view* vw, const float* pmv, const orc_brick_grid* g, const uint32_t* counters,
                             const uint8_t* mask, int px, int py, float* out)
{
  out[0] = 1.0f;
  out[1] = 0.0f;
  out[2] = 1.0f;
  out[3] = 0.0f;
  /* world-space ray: camera -> far-plane point of the pixel centre */
  const float z4[4] = {0.0f, 0.0f, 0.0f, 1.0f};
  float o4[4], pc[4], es[4], f4[4];
  mat4_mul_vec4(vw->modelview_inv, z4, o4);
  const float frag[4] = {(float)px + 0.5f, (float)py + 0.5f, 1.0f, 1.0f};
  mat4_mul_vec4(vw->img_to_eye, frag, pc);
  es[0] = pc[0] / pc[3];
  es[1] = pc[1] / pc[3];
  es[2] = pc[2] / pc[3];
  es[3] = 1.0f;
  mat4_mul_vec4(vw->modelview_inv, es, f4);
  const float o[3] = {o4[0], o4[1], o4[2]};
  const float d[3] = {f4[0] - o4[0], f4[1] - o4[1], f4[2] - o4[2]};
  /* clip the segment t in [0,1] against the brick grid's box */
  float t0 = 0.0f, t1 = 1.0f;
  int entry_axis = -1; /* the axis of the box face the ray enters the grid through (when it starts outside) */
  for (int a = 0; a < 3; ++a) {
    const float lo = g->bbox_min[a], hi = g->bbox_min[a] + g->brick_size * (float)g->res_bricks[a];
    const float inv = 1.0f / d[a];
    const float ta = (lo - o[a]) * inv, tb = (hi - o[a]) * inv;
    const float tin = fminf(ta, tb);
    if (tin > t0) {
      t0 = tin;
      entry_axis = a;
    }
    t1 = fminf(t1, fmaxf(ta, tb));
  }
  if (!(t0 < t1)) return;
  /* Amanatides-Woo walk */
  int cell[3], stepi[3];
  float tmax[3], tdelta[3];
  const float tstart = t0;
  for (int a = 0; a < 3; ++a) {
    const float pos = (o[a] + d[a] * tstart - g->bbox_min[a]) / g->brick_size;
    int c = (int)floorf(pos);
    if (c < 0) c = 0;
    if (c > g->res_bricks[a] - 1) c = g->res_bricks[a] - 1;
    cell[a] = c;
    if (d[a] > 0.0f) {
      stepi[a] = 1;
      tmax[a] = (g->bbox_min[a] + g->brick_size * (float)(c + 1) - o[a]) / d[a];
      tdelta[a] = g->brick_size / d[a];
    } else if (d[a] < 0.0f) {
      stepi[a] = -1;
      tmax[a] = (g->bbox_min[a] + g->brick_size * (float)c - o[a]) / d[a];
      tdelta[a] = -g->brick_size / d[a];
    } else {
      stepi[a] = 0;
      tmax[a] = INFINITY;
      tdelta[a] = INFINITY;
    }
  }
  float r = 1.0f, gneg = 0.0f, b = 1.0f;
  /* entry into the grid from outside (t0 > 0): previous cell is outside the grid */
  int prev_in_grid = 0, prev[3] = {cell[0], cell[1], cell[2]};
  if (entry_axis >= 0) prev[entry_axis] -= stepi[entry_axis]; /* the cell outside the grid the ray comes from */
  float tcur = t0;
  int first = 1;
  for (int iter = 0; iter < 4096; ++iter) {
    /* boundary between prev and cell at tcur (skipped for the start cell when the ray begins inside it) */
    if (!(first && !(t0 > 0.0f))) {
      const int cur_list = grid_in_list(g, mask, cell);
      const int prev_list = prev_in_grid ? grid_in_list(g, mask, prev) : 0;
      if (cur_list || prev_list) {
        const float z = peel_z(pmv, o, d, tcur);
        if (z >= 0.0f && z <= 1.0f) {
          if (cur_list && !grid_counter_gt10(g, counters, prev)) { /* front face of `cell` */
            r = fminf(r, z);
            gneg = fminf(gneg, -z);
          }
          if (prev_list && !grid_counter_gt10(g, counters, cell)) { /* back face of `prev` */
            r = fminf(r, z);
            gneg = fminf(gneg, -z);
            b = fminf(b, z);
          }
        }
      }
    }
    first = 0;
    /* advance to the next cell */
    int a = 0;
    if (tmax[1] < tmax[a]) a = 1;
    if (tmax[2] < tmax[a]) a = 2;
    const float tnext = tmax[a];
    if (!(tnext < t1)) {
      /* leaving through the grid's outer box at t1: back face of the last cell */
      int beyond[3] = {cell[0], cell[1], cell[2]};
      beyond[a] += stepi[a];
      if (grid_in_list(g, mask, cell) && !grid_counter_gt10(g, counters, beyond)) {
        const float z = peel_z(pmv, o, d, t1);
        if (t1 < 1.0f && z >= 0.0f && z <= 1.0f) {
          r = fminf(r, z);
          gneg = fminf(gneg, -z);
          b = fminf(b, z);
        }
      }
      break;
    }
    prev[0] = cell[0];
    prev[1] = cell[1];
    prev[2] = cell[2];
    prev_in_grid = 1;
    cell[a] += stepi[a];
    tmax[a] += tdelta[a];
    tcur = tnext;
    if (cell[a] < 0 || cell[a] >= g->res_bricks[a]) {
      /* stepped out of the grid: back face of prev */
      if (grid_in_list(g, mask, prev) && !grid_counter_gt10(g, counters, cell)) {
        const float z = peel_z(pmv, o, d, tcur);
        if (z >= 0.0f && z <= 1.0f) {
          r = fminf(r, z);
          gneg = fminf(gneg, -z);
          b = fminf(b, z);
        }
      }
      break;
    }
  }
  out[0] = r;
  out[1] = gneg;
  out[2] = b;
  out[3] = 0.0f;
}

static void mat4_mul_mat4(const float* a, const float* bm, float* o)
{
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r)
      o[4 * c + r] = a[r] * bm[4 * c] + a[4 + r] * bm[4 * c + 1] + a[8 + r] * bm[4 * c + 2] + a[12 + r] * bm[4 * c + 3];
}

ORC_API void orc_depth_peels(const orc_view* vw, const orc_brick_grid* g, const uint32_t* counters,
                             const uint8_t* mask, float* out /* H*W*4 */)
{
  float pmv[16];
  mat4_mul_mat4(vw->projection, vw->modelview, pmv); /* gl_ProjectionMatrix * gl_ModelViewMatrix, bricks.vs:19 */
#pragma omp parallel for schedule(dynamic, 4)
  for (int py = 0; py < vw->height; ++py)
    for (int px = 0; px < vw->width; ++px)
      depth_peel_pixel(vw, pmv, g, counters, mask, px, py, out + ((size_t)py * vw->width + px) * 4);
}

static inline float tsdf_sample(const float* tsdf, const int* res, const float* p)
{
  float o;
  tex3d_linear(tsdf, 1, 1, res[0], res[1], res[2], p[0], p[1], p[2], &o);
  return o;
}

static void rm_gradient(const float* tsdf, const int* res, const float* pos, float sd, float* g)
{
  float d[3];
  for (int a = 0; a < 3; ++a) {
    float pp[3] = {pos[0], pos[1], pos[2]}, pm[3] = {pos[0], pos[1], pos[2]};
    pp[a] = pos[a] + sd;
    pm[a] = pos[a] - sd;
    d[a] = tsdf_sample(tsdf, res, pp) - tsdf_sample(tsdf, res, pm);
  }
  float n[3];
  normalize3(d, n);
  g[0] = -n[0];
  g[1] = -n[1];
  g[2] = -n[2];
}

static const float rm_camera_colors[5][3] = {{228, 26, 28}, {55, 126, 184}, {77, 175, 74}, {152, 78, 163}, {255, 127, 0}};

ORC_API void orc_raymarch(const orc_view* vw, const orc_raymarch_params* p, const float* tsdf /* Z*Y*X */,
                          const float* const* cv_xyz_inv /* RGBA */, const int* inv_res, const float* const* cv_uv,
                          const int* uv_res, const uint8_t* const* colors, const float* const* depth_b_rg,
                          const float* const* quality, const float* peels /* H*W*4 when skip_space */,
                          float* out_color /* H*W*4 */, float* out_depth /* H*W */, float* out_samples /* H*W */)
{
  const float limit = p->limit, sd = limit * 0.5f;
  /* gl_ModelViewMatrix * vol_to_world (the shader's left-to-right product, :123), glm association */
  float mvw[16];
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r)
      mvw[4 * c + r] = vw->modelview[r] * vw->vol_to_world[4 * c] + vw->modelview[4 + r] * vw->vol_to_world[4 * c + 1] +
                       vw->modelview[8 + r] * vw->vol_to_world[4 * c + 2] + vw->modelview[12 + r] * vw->vol_to_world[4 * c + 3];
#pragma omp parallel for schedule(dynamic, 4)
  for (int py = 0; py < vw->height; ++py) {
    for (int px = 0; px < vw->width; ++px) {
      const size_t o = (size_t)py * vw->width + px;
      /* cleared framebuffer: ViewLod::enable, glClearColor(0,1,0,0), depth 1 */
      out_color[o * 4 + 0] = 0.0f;
      out_color[o * 4 + 1] = 1.0f;
      out_color[o * 4 + 2] = 0.0f;
      out_color[o * 4 + 3] = 0.0f;
      out_depth[o] = 1.0f;
      out_samples[o] = 0.0f;
      /* ray through the pixel centre: screenToVol of the far-plane point */
      const float frag[4] = {(float)px + 0.5f, (float)py + 0.5f, 1.0f, 1.0f};
      float pc[4], es[4], ws[4], tv[4];
      mat4_mul_vec4(vw->img_to_eye, frag, pc);
      es[0] = pc[0] / pc[3];
      es[1] = pc[1] / pc[3];
      es[2] = pc[2] / pc[3];
      es[3] = 1.0f;
      mat4_mul_vec4(vw->modelview_inv, es, ws);
      mat4_mul_vec4(vw->vol_to_world_inv, ws, tv);
      float dir[3] = {tv[0] - vw->camera_pos[0], tv[1] - vw->camera_pos[1], tv[2] - vw->camera_pos[2]}, nd[3];
      normalize3(dir, nd);
      const float step[3] = {nd[0] * sd, nd[1] * sd, nd[2] * sd};
      /* intersectBox(CameraPos, sampleStep), :371-382 */
      float tmin[3], tmax[3];
      for (int a = 0; a < 3; ++a) {
        const float inv = 1.0f / step[a];
        const float tb = inv * (0.0f - vw->camera_pos[a]), tt = inv * (1.0f - vw->camera_pos[a]);
        tmin[a] = fminf(tt, tb);
        tmax[a] = fmaxf(tt, tb);
      }
      const float t0 = fmaxf(fmaxf(tmin[0], tmin[1]), fmaxf(tmin[0], tmin[2]));
      const float t1 = fminf(fminf(tmax[0], tmax[1]), fminf(tmax[0], tmax[2]));
      const int is_t0 = t0 <= t1;
      if (!is_t0 || !(t1 > 0.0f)) continue; /* the cube is not rasterised onto this pixel */
      float t_near = is_t0 ? t0 : t1;
      t_near = t_near < 0.0f ? 0.0f : t_near;
      const float t_far = is_t0 ? t1 : t0;
      float sp[3] = {vw->camera_pos[0] + step[0] * t_near, vw->camera_pos[1] + step[1] * t_near,
                     vw->camera_pos[2] + step[2] * t_near};
      float fmaxs = ceilf(fabsf(t_far - t_near));
      if (vw->skip_space) { /* getStartPos, :392-401 */
        float dr = peels[o * 4 + 0];
        const float dg = peels[o * 4 + 1], dbk = peels[o * 4 + 2];
        dr = (dr >= dbk) ? 0.0f : dr; /* gl_DepthRange.near */
        float pf[3], pb[3];
        for (int k = 0; k < 2; ++k) {
          const float fr[4] = {(float)px + 0.5f, (float)py + 0.5f, k == 0 ? dr : -dg, 1.0f};
          float a4[4], e4[4], w4[4], v4[4];
          mat4_mul_vec4(vw->img_to_eye, fr, a4);
          e4[0] = a4[0] / a4[3];
          e4[1] = a4[1] / a4[3];
          e4[2] = a4[2] / a4[3];
          e4[3] = 1.0f;
          mat4_mul_vec4(vw->modelview_inv, e4, w4);
          mat4_mul_vec4(vw->vol_to_world_inv, w4, v4);
          float* dst = k == 0 ? pf : pb;
          dst[0] = v4[0];
          dst[1] = v4[1];
          dst[2] = v4[2];
        }
        if (dr >= 1.0f) {
          pb[0] = pf[0];
          pb[1] = pf[1];
          pb[2] = pf[2];
        }
        sp[0] = pf[0];
        sp[1] = pf[1];
        sp[2] = pf[2];
        fmaxs = ceilf(distance3(pf, pb) / sd);
      }
      const unsigned max_num = !(fmaxs > 0.0f) ? 0u : (fmaxs >= 2147483520.0f ? 2147483520u : (unsigned)fmaxs);
      float prev = -limit;
      unsigned num = 0;
      int hit = 0;
      while (num < max_num) {
        num += 1u;
        const float density = tsdf_sample(tsdf, p->res, sp);
        if (density > 0.0f) {
          const float f = prev / (density - prev);
          for (int a = 0; a < 3; ++a) sp[a] = (sp[a] - step[a]) - step[a] * f;
          hit = 1;
          break;
        }
        prev = density;
        for (int a = 0; a < 3; ++a) sp[a] += step[a];
      }
      out_samples[o] = (float)num * 0.0027f;
      if (!hit) continue; /* discard */
      /* submitFragment, :116-142 */
      float g[3], g4[4], vn4[4], vn[3];
      rm_gradient(tsdf, p->res, sp, sd, g);
      g4[0] = g[0];
      g4[1] = g[1];
      g4[2] = g[2];
      g4[3] = 0.0f;
      mat4_mul_vec4(vw->normal_matrix, g4, vn4);
      normalize3(vn4, vn);
      const float sp4[4] = {sp[0], sp[1], sp[2], 1.0f};
      float vp[4];
      mat4_mul_vec4(mvw, sp4, vp);
      float rgba[4];
      /* per-sensor lookups shared by blendColors / blendCameras */
      float tc[3] = {0, 0, 0}, tc2[3] = {0, 0, 0}, tw = 0.0f, tw2 = 0.0f, cw[3] = {0, 0, 0}, cwt = 0.0f;
      for (int i = 0; i < p->num_sensors; ++i) {
        float pcal[3], pcol[2], col[3];
        tex3d_linear(cv_xyz_inv[i], 4, 3, inv_res[3 * i], inv_res[3 * i + 1], inv_res[3 * i + 2], sp[0], sp[1], sp[2], pcal);
        tex3d_linear(cv_uv[i], 2, 2, uv_res[3 * i], uv_res[3 * i + 1], uv_res[3 * i + 2], pcal[0], pcal[1], pcal[2], pcol);
        tex2d_linear_rgb8(colors[i], p->Wc, p->Hc, pcol[0], pcol[1], col);
        const int ix = axis_nearest(pcal[0], p->W), iy = axis_nearest(pcal[1], p->H);
        const float depth = depth_b_rg[i][((size_t)iy * p->W + ix) * 2];
        const float dist = fabsf(depth - pcal[2]);
        float q = 0.0f;
        if (dist < limit) tex2d_linear(quality[i], 1, 1, p->W, p->H, pcal[0], pcal[1], &q);
        for (int k = 0; k < 3; ++k) {
          tc[k] += col[k] * q / (dist + 0.01f);
          tc2[k] += col[k] / dist;
          if (i < 5) cw[k] += (rm_camera_colors[i][k] / 255.0f) * q;
        }
        tw += q / (dist + 0.01f);
        tw2 += 1.0f / dist;
        cwt += q;
      }
      if (vw->shade_mode == 3) { /* blendCameras, :354-369 */
        for (int k = 0; k < 3; ++k) rgba[k] = (cwt <= 0.0f) ? 1.0f : cw[k] / cwt;
        rgba[3] = 1.0f;
      } else {
        float diff[4];
        if (tw > 0.0f) {
          for (int k = 0; k < 3; ++k) diff[k] = tc[k] / tw;
          diff[3] = 1.0f;
        } else {
          for (int k = 0; k < 3; ++k) diff[k] = tc2[k] / tw2;
          diff[3] = -1.0f;
        }
        if (vw->shade_mode == 0) {
          rgba[0] = diff[0];
          rgba[1] = diff[1];
          rgba[2] = diff[2];
        } else if (vw->shade_mode == 1) { /* phong, shading.glsl:32-64 */
          const float lp[3] = {1.5f, 1.0f, 1.0f};
          float tl[3] = {lp[0] - vp[0], lp[1] - vp[1], lp[2] - vp[2]}, tln[3];
          normalize3(tl, tln);
          const float la = dot3(vn, tln);
          float dc = 0.0f, sl = 0.0f;
          if (!(la <= 0.0f)) {
            dc = fmaxf(la, 0.0f);
            const float nv[3] = {-vp[0], -vp[1], -vp[2]};
            float tvw[3], hw[3], hn[3];
            normalize3(nv, tvw);
            hw[0] = tln[0] + tvw[0];
            hw[1] = tln[1] + tvw[1];
            hw[2] = tln[2] + tvw[2];
            normalize3(hw, hn);
            const float ra = dot3(hn, vn);
            /* pow(reflectedAngle, 20): products, like the other constant exponents */
            const float r2 = ra * ra, r4 = r2 * r2, r8 = r4 * r4, r16 = r8 * r8;
            sl = r16 * r4;
            const float a = (1.0f - la) * (1.0f - la);
            sl *= 1.0f - a * a * a;
          }
          const float ld[3] = {1.0f, 0.9f, 0.7f};
          for (int k = 0; k < 3; ++k) rgba[k] = (ld[k] * 0.2f) * 0.5f + ld[k] * 0.5f * dc + 1.0f * 0.5f * sl;
        } else if (vw->shade_mode == 2) {
          const float v4[4] = {vn[0], vn[1], vn[2], 0.0f};
          float r4[4];
          mat4_mul_vec4(vw->gl_normal_matrix_inv, v4, r4);
          rgba[0] = r4[0];
          rgba[1] = r4[1];
          rgba[2] = r4[2];
        } else {
          rgba[0] = rgba[1] = rgba[2] = 1.0f;
        }
        rgba[3] = diff[3];
      }
      /* gl_FragDepth, :133 (projection[2].z = m[10], projection[3].z = m[14]); it is clamped
       * to the depth range and tested GL_LESS against the cleared 1.0
       * (source/kinect_client.cpp:993-994), so a surface at or beyond the far plane is dropped */
      const float fd = (vw->projection[10] * vp[2] + vw->projection[14]) / -vp[2] * 0.5f + 0.5f;
      const float dclamped = fminf(fmaxf(fd, 0.0f), 1.0f);
      if (!(dclamped < 1.0f)) continue;
      for (int k = 0; k < 4; ++k) out_color[o * 4 + k] = rgba[k];
      out_depth[o] = dclamped;
    }
  }
}

/* ------------------------------------------------------------------------- */
/* Hole filling of the ray-marched frame (f-2): ReconIntegration::fillColors    */
/* (recon_integration.cpp:280-339) with glsl/framebuffer_transfer.fs,            */
/* glsl/tsdf_inpaint.fs, glsl/tsdf_colorfill.fs and the LOD atlas of ViewLod      */
/* (framework/rendering/view_lod.cpp:24-61: 1.5*W x H, LOD i at (W, H - sum h_j)). */
/* Two atlases ping-pong in the reference: the "native" one accumulates the LODs, */
/* the other holds a copy of it squeezed by 2/3 in x (framebuffer_transfer.fs     */
/* fetches at pass_TexCoord * resolution_full).  Conventions where GL leaves the  */
/* result open: texelFetch outside the texture returns 0; uniform array entries   */
/* beyond num_lods are 0; clamp(x, lo, hi) = min(max(x, lo), hi) also for lo > hi; */
/* the colour atlas is sampled LINEAR with MIRRORED_REPEAT (view_lod.cpp:52-53).   */

typedef struct {
  int W, H, FW, num_lods;
  int off[20][2], res[20][2];
} fc_layout;

static void fc_make_layout(int W, int H, fc_layout* L)
{
  memset(L, 0, sizeof(*L));
  L->W = W;
  L->H = H;
  L->FW = (int)((float)W * 1.5f);
  int m = W < H ? W : H;
  L->num_lods = 1 + (int)floorf(log2f((float)m));
  if (L->num_lods > 20) L->num_lods = 20;
  int ox = W, oy = H;
  for (int i = 0; i < L->num_lods; ++i) {
    L->res[i][0] = (int)floorf((float)W / powf(2.0f, (float)i));
    L->res[i][1] = (int)floorf((float)H / powf(2.0f, (float)i));
    if (i > 0) {
      oy -= L->res[i][1];
      L->off[i][0] = ox;
      L->off[i][1] = oy;
    }
  }
}

ORC_API void orc_fill_layout(int W, int H, int* num_lods, int* full_w, int* off /* 20x2 */, int* res /* 20x2 */)
{
  fc_layout L;
  fc_make_layout(W, H, &L);
  *num_lods = L.num_lods;
  *full_w = L.FW;
  memcpy(off, L.off, sizeof(L.off));
  memcpy(res, L.res, sizeof(L.res));
}

static inline void fc_fetch(const float* col, const float* dep, const fc_layout* L, int x, int y, float* c, float* d)
{
  if (x < 0 || y < 0 || x >= L->FW || y >= L->H) {
    c[0] = c[1] = c[2] = c[3] = 0.0f;
    *d = 0.0f;
    return;
  }
  memcpy(c, col + ((size_t)y * L->FW + x) * 4, 16);
  *d = dep[(size_t)y * L->FW + x];
}

static inline int fc_mirror(int i, int n)
{
  int period = 2 * n;
  int k = i % period;
  if (k < 0) k += period;
  return k < n ? k : period - 1 - k;
}

/* texture(texture_color, p): LINEAR, MIRRORED_REPEAT, p normalised over the atlas */
static void fc_texture(const float* col, const fc_layout* L, float u, float v, float* out)
{
  const float tx = u * (float)L->FW - 0.5f, ty = v * (float)L->H - 0.5f;
  const float fx = floorf(tx), fy = floorf(ty);
  const float ax = tx - fx, ay = ty - fy;
  const int jx = idx_from_floor(fx, L->FW * 4), jy = idx_from_floor(fy, L->H * 4);
  const int x0 = fc_mirror(jx, L->FW), x1 = fc_mirror(jx + 1, L->FW), y0 = fc_mirror(jy, L->H), y1 = fc_mirror(jy + 1, L->H);
  for (int c = 0; c < 4; ++c) {
    const float t00 = col[((size_t)y0 * L->FW + x0) * 4 + c], t10 = col[((size_t)y0 * L->FW + x1) * 4 + c];
    const float t01 = col[((size_t)y1 * L->FW + x0) * 4 + c], t11 = col[((size_t)y1 * L->FW + x1) * 4 + c];
    out[c] = lerpf(lerpf(t00, t10, ax), lerpf(t01, t11, ax), ay);
  }
}

/* the sampler above for the shader-text harness (oracle/glsl_runtime.hpp): an FW x H RGBA32F texture */
ORC_API void orc_tex2d_linear_mirrored_rgba(const float* col, int FW, int H, float u, float v, float* out)
{
  fc_layout L;
  memset(&L, 0, sizeof(L));
  L.FW = FW;
  L.H = H;
  fc_texture(col, &L, u, v, out);
}

static void fc_clear(float* col, float* dep, const fc_layout* L)
{
  for (size_t i = 0; i < (size_t)L->FW * L->H; ++i) {
    col[4 * i] = 0.0f;
    col[4 * i + 1] = 1.0f;
    col[4 * i + 2] = 0.0f;
    col[4 * i + 3] = 0.0f;
    dep[i] = 1.0f;
  }
}

/* framebuffer_transfer.fs into the LOD-0 viewport of a freshly cleared atlas */
static void fc_transfer(const float* scol, const float* sdep, float* dcol, float* ddep, const fc_layout* L)
{
  fc_clear(dcol, ddep, L);
  for (int py = 0; py < L->H; ++py)
    for (int px = 0; px < L->W; ++px) {
      const float u = ((float)px + 0.5f) / (float)L->W, v = ((float)py + 0.5f) / (float)L->H;
      const int sx = (int)(u * (float)L->FW), sy = (int)(v * (float)L->H);
      fc_fetch(scol, sdep, L, sx, sy, dcol + ((size_t)py * L->FW + px) * 4, ddep + (size_t)py * L->FW + px);
    }
}

/* tsdf_inpaint.fs: reads the squeezed atlas, writes LOD `lod + 1` of the native one */
static void fc_inpaint(const float* scol, const float* sdep, float* ncol, float* ndep, const fc_layout* L, int lod)
{
  const int i = lod + 1;
  for (int fy = 0; fy < L->res[i][1]; ++fy)
    for (int fx = 0; fx < L->res[i][0]; ++fx) {
      const int gx = L->off[i][0] + fx, gy = L->off[i][1] + fy; /* gl_FragCoord, pixel_center_integer */
      const float tcx = ((float)gx - (float)L->off[i][0]) / (float)L->res[i][0];
      const float tcy = ((float)gy - (float)L->off[i][1]) / (float)L->res[i][1];
      const int lx = (int)((float)L->off[lod][0] + (float)L->res[lod][0] * tcx);
      const int ly = (int)((float)L->off[lod][1] + (float)L->res[lod][1] * tcy);
      const int pix = (int)((float)lx * (2.0f / 3.0f)), piy = (int)((float)ly * 1.0f);
      float samples[16][4], depth_av = 0.0f;
      int num = 0;
      for (int x = 0; x < 4; ++x)
        for (int y = 0; y < 4; ++y) {
          const int tx = pix + (int)((float)x - 4.0f * 0.5f + 1.0f), ty = piy + (int)((float)y - 4.0f * 0.5f + 1.0f);
          float c[4], d;
          fc_fetch(scol, sdep, L, tx, ty, c, &d);
          if (c[3] <= 0.0f) {
            c[0] = -1.0f;
          } else {
            depth_av += d;
            ++num;
          }
          float* sm = samples[x + y * 4];
          sm[0] = c[0];
          sm[1] = c[1];
          sm[2] = c[2];
          sm[3] = d;
        }
      float* oc = ncol + ((size_t)gy * L->FW + gx) * 4;
      float* od = ndep + (size_t)gy * L->FW + gx;
      if (num == 0) {
        float c[4], d;
        fc_fetch(scol, sdep, L, pix, piy, c, &d);
        *od = d;
        if (d < 1.0f) {
          oc[0] = 0.0f;
          oc[1] = 0.0f;
          oc[2] = 0.0f;
          oc[3] = -1.0f;
        } else {
          oc[0] = 0.0f;
          oc[1] = 1.0f;
          oc[2] = 0.0f;
          oc[3] = 0.0f;
        }
        continue;
      }
      depth_av /= (float)num;
      float tc[3] = {0, 0, 0}, td = 0.0f, tw = 0.0f;
      for (int k = 0; k < 16; ++k)
        if (samples[k][0] >= 0.0f && samples[k][3] >= depth_av) {
          tc[0] += samples[k][0] * 1.0f;
          tc[1] += samples[k][1] * 1.0f;
          tc[2] += samples[k][2] * 1.0f;
          td += samples[k][3] * 1.0f;
          tw += 1.0f;
        }
      oc[0] = tc[0] / tw;
      oc[1] = tc[1] / tw;
      oc[2] = tc[2] / tw;
      oc[3] = 1.0f;
      *od = td / tw;
    }
}

static inline float fc_clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

/* tsdf_colorfill.fs into the W x H default framebuffer */
static void fc_colorfill(const float* ncol, const float* ndep, const fc_layout* L, float* out_col, float* out_dep)
{
  const float rix = 1.0f / (float)L->FW, riy = 1.0f / (float)L->H; /* resolution_inv */
  for (int py = 0; py < L->H; ++py)
    for (int px = 0; px < L->W; ++px) {
      const float tcx = (float)px / (float)L->res[0][0], tcy = (float)py / (float)L->res[0][1];
      const float ptx = ((float)px + 0.5f) / (float)L->W, pty = ((float)py + 0.5f) / (float)L->H; /* pass_TexCoord */
      float c[4] = {0, 0, 0, 0}, d;
      int level = 0;
      for (; level < L->num_lods; ++level) {
        const int cx = (int)((float)L->off[level][0] + (float)L->res[level][0] * tcx);
        const int cy = (int)((float)L->off[level][1] + (float)L->res[level][1] * tcy);
        fc_fetch(ncol, ndep, L, cx, cy, c, &d);
        if (c[3] > 0.0f) break;
      }
      if (level > 0) {
        float p[2][2];
        for (int k = 0; k < 2; ++k) {
          const int l = level + 1 + k; /* level+1 -> p1, level+2 -> p2 */
          const float ox = l < 20 ? (float)L->off[l][0] : 0.0f, oy = l < 20 ? (float)L->off[l][1] : 0.0f;
          const float rx = l < 20 ? (float)L->res[l][0] : 0.0f, ry = l < 20 ? (float)L->res[l][1] : 0.0f;
          p[k][0] = fc_clampf(ox + rx * ptx, ox + 0.5f, (ox + rx) - 0.5f) * rix;
          p[k][1] = fc_clampf(oy + ry * pty, oy + 0.5f, (oy + ry) - 0.5f) * riy;
        }
        float c1[4], c2[4];
        fc_texture(ncol, L, p[0][0], p[0][1], c1);
        fc_texture(ncol, L, p[1][0], p[1][1], c2);
        const float w1 = sqrtf(ptx * ptx + pty * pty); /* distance(pass_TexCoord, floor(pass_TexCoord)) */
        const float w2 = 1.0f - w1;
        for (int k = 0; k < 4; ++k) c[k] = (c1[k] * w1 + c2[k] * w2) / (w1 + w2);
      }
      float c0[4], d0;
      fc_fetch(ncol, ndep, L, (int)((float)L->off[0][0] + (float)L->res[0][0] * tcx),
               (int)((float)L->off[0][1] + (float)L->res[0][1] * tcy), c0, &d0);
      memcpy(out_col + ((size_t)py * L->W + px) * 4, c, 16);
      out_dep[(size_t)py * L->W + px] = d0;
    }
}

ORC_API void orc_fill_colors(const float* color /* H*W*4 */, const float* depth /* H*W */, int W, int H,
                             float* out_color, float* out_depth, float* atlas_color /* H*FW*4 or NULL */)
{
  fc_layout L;
  fc_make_layout(W, H, &L);
  const size_t n = (size_t)L.FW * H;
  float* ncol = (float*)malloc(n * 16);
  float* ndep = (float*)malloc(n * 4);
  float* scol = (float*)malloc(n * 16);
  float* sdep = (float*)malloc(n * 4);
  /* m_view_inpaint after draw(): cleared atlas with the ray-marched frame in LOD 0 */
  fc_clear(ncol, ndep, &L);
  for (int y = 0; y < H; ++y) {
    memcpy(ncol + (size_t)y * L.FW * 4, color + (size_t)y * W * 4, (size_t)W * 16);
    memcpy(ndep + (size_t)y * L.FW, depth + (size_t)y * W, (size_t)W * 4);
  }
  fc_transfer(ncol, ndep, scol, sdep, &L);
  for (int i = 1; i < L.num_lods; ++i) {
    fc_inpaint(scol, sdep, ncol, ndep, &L, i - 1);
    fc_transfer(ncol, ndep, scol, sdep, &L);
  }
  fc_colorfill(ncol, ndep, &L, out_color, out_depth);
  if (atlas_color) memcpy(atlas_color, ncol, n * 16);
  free(ncol);
  free(ndep);
  free(scol);
  free(sdep);
}

/* ------------------------------------------------------------------------- */
/* Frustum planes + inside test (framework/calibration/frustum.cpp:36-43,      */
/* :113-177) and the offline inverter CalibrationInverter::calculateInverse-    */
/* Volumes (framework/calibration/calibration_inverter.cpp:55-69, :99-155).     */
/* The reference finds the 8 nearest cv_xyz samples with a CGAL k-d tree (not    */
/* available here, version unpinned: "parity unpinned"); this restatement does   */
/* an exact brute-force search, orders the neighbours by ascending squared       */
/* distance (ties: lower sample index in getXyzSamples order, x outer / z inner) */
/* and accumulates the inverse-distance weights in that order.                   */

static void corner_points(const float* cv_xyz, const int* res, float c[8][3])
{
  const int ex = res[0] - 1, ey = res[1] - 1, ez = res[2] - 1;
  const int cx[8] = {0, 0, ex, ex, 0, 0, ex, ex};
  const int cy[8] = {0, ey, ey, 0, 0, ey, ey, 0};
  const int cz[8] = {0, 0, 0, 0, ez, ez, ez, ez};
  for (int i = 0; i < 8; ++i)
    for (int k = 0; k < 3; ++k) c[i][k] = cv_xyz[(((size_t)cz[i] * res[1] + cy[i]) * res[0] + cx[i]) * 3 + k];
}

ORC_API void orc_frustum_planes(const float* cv_xyz, const int* res, float* planes /* 6 x 4 */)
{
  float c[8][3], e[12][3], sc[6][3], n[6][3];
  corner_points(cv_xyz, res, c);
  const int ea[12] = {0, 1, 2, 3, 4, 5, 6, 7, 0, 1, 2, 3}, eb[12] = {1, 2, 3, 0, 5, 6, 7, 4, 4, 5, 6, 7};
  for (int i = 0; i < 12; ++i)
    for (int k = 0; k < 3; ++k) e[i][k] = (c[ea[i]][k] + c[eb[i]][k]) * 0.5f;
  /* getSideCenters: near, far, left, right, top, bottom */
  const int sq[6][4] = {{0, 1, 2, 3}, {4, 5, 6, 7}, {0, 1, 4, 5}, {2, 3, 6, 7}, {1, 2, 5, 6}, {0, 3, 4, 7}};
  for (int i = 0; i < 6; ++i)
    for (int k = 0; k < 3; ++k) sc[i][k] = (c[sq[i][0]][k] + c[sq[i][1]][k] + c[sq[i][2]][k] + c[sq[i][3]][k]) / 4.0f;
  /* getSideNormals: cross(e[a]-e[b], e[c]-e[d]) normalised */
  const int nq[6][4] = {{0, 2, 3, 2}, {4, 6, 5, 7}, {0, 4, 9, 8}, {2, 6, 11, 10}, {9, 10, 1, 5}, {8, 11, 7, 3}};
  for (int i = 0; i < 6; ++i) {
    float a[3], b[3], x[3];
    for (int k = 0; k < 3; ++k) {
      a[k] = e[nq[i][0]][k] - e[nq[i][1]][k];
      b[k] = e[nq[i][2]][k] - e[nq[i][3]][k];
    }
    cross3(a, b, x);
    normalize3(x, n[i]);
    planes[4 * i + 0] = n[i][0];
    planes[4 * i + 1] = n[i][1];
    planes[4 * i + 2] = n[i][2];
    planes[4 * i + 3] = -dot3(n[i], sc[i]);
  }
}

static inline int frustum_inside(const float* planes, const float* p)
{
  for (int i = 0; i < 6; ++i) {
    /* glm::dot(vec4, vec4): (x*x + y*y) + (z*z + w*w) */
    const float d = (planes[4 * i] * p[0] + planes[4 * i + 1] * p[1]) + (planes[4 * i + 2] * p[2] + planes[4 * i + 3] * 1.0f);
    if (d < 0.0f) return 0;
  }
  return 1;
}

ORC_API int orc_frustum_inside(const float* planes, const float* p) { return frustum_inside(planes, p); }

ORC_API void orc_inverse_volume(const float* cv_xyz, const int* res, const float* bbox_min, const float* bbox_max,
                                const int* vol_res, int z0, int z1, float* out /* RGBA, rows [z0,z1) */)
{
  float planes[24];
  orc_frustum_planes(cv_xyz, res, planes);
  const int rx = res[0], ry = res[1], rz = res[2];
  float step[3], start[3];
  for (int a = 0; a < 3; ++a) {
    const float vstep = 1.0f / (float)vol_res[a];
    step[a] = (bbox_max[a] - bbox_min[a]) * vstep;
    start[a] = bbox_min[a] + step[a] * 0.5f;
  }
#pragma omp parallel for collapse(2) schedule(dynamic, 8)
  for (int z = z0; z < z1; ++z) {
    for (int y = 0; y < vol_res[1]; ++y) {
      for (int x = 0; x < vol_res[0]; ++x) {
        float* o = out + (((size_t)(z - z0) * vol_res[1] + y) * vol_res[0] + x) * 4;
        const float p[3] = {start[0] + (float)x * step[0], start[1] + (float)y * step[1], start[2] + (float)z * step[2]};
        if (!frustum_inside(planes, p)) {
          o[0] = o[1] = o[2] = o[3] = -1.0f;
          continue;
        }
        float bd[8];
        int bi[8], bxyz[8][3], n = 0;
        for (int sx = 0; sx < rx; ++sx)
          for (int sy = 0; sy < ry; ++sy)
            for (int sz = 0; sz < rz; ++sz) {
              const float* sp = cv_xyz + (((size_t)sz * ry + sy) * rx + sx) * 3;
              const float d[3] = {p[0] - sp[0], p[1] - sp[1], p[2] - sp[2]};
              const float d2 = dot3(d, d);
              const int lin = (sx * ry + sy) * rz + sz;
              if (n == 8 && !(d2 < bd[7] || (d2 == bd[7] && lin < bi[7]))) continue;
              int k = n < 8 ? n : 7;
              while (k > 0 && (d2 < bd[k - 1] || (d2 == bd[k - 1] && lin < bi[k - 1]))) {
                bd[k] = bd[k - 1];
                bi[k] = bi[k - 1];
                memcpy(bxyz[k], bxyz[k - 1], sizeof(bxyz[0]));
                --k;
              }
              bd[k] = d2;
              bi[k] = lin;
              bxyz[k][0] = sx;
              bxyz[k][1] = sy;
              bxyz[k][2] = sz;
              if (n < 8) ++n;
            }
        float tw = 0.0f, wi[3] = {0.0f, 0.0f, 0.0f};
        for (int k = 0; k < n; ++k) {
          const float w = 1.0f / sqrtf(bd[k]);
          wi[0] += w * (float)bxyz[k][0];
          wi[1] += w * (float)bxyz[k][1];
          wi[2] += w * (float)bxyz[k][2];
          tw += w;
        }
        o[0] = (wi[0] / tw + 0.5f) / (float)rx;
        o[1] = (wi[1] / tw + 0.5f) / (float)ry;
        o[2] = (wi[2] / tw + 0.5f) / (float)rz;
        o[3] = 1.0f;
      }
    }
  }
}

/* ------------------------------------------------------------------------- */
/* Grid geometry (a8): setVoxelSize / setBrickSize / divideBox                */
/* framework/reconstruction/recon_integration.cpp:341-354, :474-484, :361-388 */

ORC_API void orc_volume_res(const float* bbox_min, const float* bbox_max, float voxel_size, int* res)
{
  for (int a = 0; a < 3; ++a) res[a] = (int)ceilf((bbox_max[a] - bbox_min[a]) / voxel_size);
}

ORC_API float orc_adjust_brick_size(float size, float voxel_size)
{
  return voxel_size * roundf(size / voxel_size);
}

/* the three nested while loops of divideBox, float accumulation included */
ORC_API void orc_divide_box(const float* bbox_min, const float* bbox_max, float brick_size, int* res_bricks)
{
  for (int a = 0; a < 3; ++a) {
    float min = bbox_min[a];
    float size = bbox_max[a] - min;
    float start = min;
    int n = 0;
    while (size - start + min > 0.0f) {
      start += brick_size;
      ++n;
    }
    res_bricks[a] = n;
  }
}

/* updateOccupiedBricks CPU filter loop (recon_integration.cpp:436-441) */
ORC_API uint32_t orc_update_occupied(const uint32_t* counters, uint32_t n, uint32_t min_voxels,
                                     uint32_t* ids, float* ratio)
{
  uint32_t k = 0;
  for (uint32_t i = 0; i < n; ++i)
    if (counters[i] >= min_voxels) ids[k++] = i;
  *ratio = (float)k / (float)n;
  return k;
}

/* Reference's genuinely-CPU work per resize:
 * VolumeSampler::resize position fill (volume_sampler.cpp:33-46).  Used by
 * bench.py only to time that loop; returns a checksum so it is not elided. */
ORC_API double orc_volume_sampler_resize(int X, int Y, int Z, float* pos /* X*Y*Z*3 */)
{
  float sx = 1.0f / X, sy = 1.0f / Y, sz = 1.0f / Z;
  size_t k = 0;
  double acc = 0.0;
  for (int z = 0; z < Z; ++z)
    for (int y = 0; y < Y; ++y)
      for (int x = 0; x < X; ++x) {
        pos[k++] = (x + 0.5f) * sx;
        pos[k++] = (y + 0.5f) * sy;
        pos[k++] = (z + 0.5f) * sz;
      }
  acc = pos[0] + pos[k - 1];
  return acc;
}

/* divideBox + VolumeSampler::containedVoxels (recon_integration.cpp:361-388,
 * volume_sampler.cpp:50-62): the per-brick voxel index vectors the reference builds
 * on the CPU at every resize (4 B per voxel), y / x / z loop order and the
 * float -> unsigned truncation of pos / step included.  bench.py times it; returns
 * the number of indices and a checksum. */
ORC_API double orc_contained_voxels(const float* bbox_min, const float* bbox_max, float brick_size, const int* dims,
                                    uint32_t* indices /* capacity >= 2 * X*Y*Z */, size_t capacity, size_t* count)
{
  const float size[3] = {bbox_max[0] - bbox_min[0], bbox_max[1] - bbox_min[1], bbox_max[2] - bbox_min[2]};
  const float stepv[3] = {1.0f / (float)dims[0], 1.0f / (float)dims[1], 1.0f / (float)dims[2]};
  size_t k = 0;
  double acc = 0.0;
  float start[3] = {bbox_min[0], bbox_min[1], bbox_min[2]};
  while (size[2] - start[2] + bbox_min[2] > 0.0f) {
    while (size[1] - start[1] + bbox_min[1] > 0.0f) {
      while (size[0] - start[0] + bbox_min[0] > 0.0f) {
        float pos[3], bs[3];
        for (int a = 0; a < 3; ++a) {
          bs[a] = fminf(brick_size, size[a] - start[a] + bbox_min[a]) / size[a];
          pos[a] = (start[a] - bbox_min[a]) / size[a];
        }
        for (unsigned y = (unsigned)(pos[1] / stepv[1]); (float)y < (pos[1] + bs[1]) / stepv[1]; ++y)
          for (unsigned x = (unsigned)(pos[0] / stepv[0]); (float)x < (pos[0] + bs[0]) / stepv[0]; ++x)
            for (unsigned z = (unsigned)(pos[2] / stepv[2]); (float)z < (pos[2] + bs[2]) / stepv[2]; ++z) {
              const uint32_t id = z * (unsigned)dims[0] * (unsigned)dims[1] + y * (unsigned)dims[0] + x;
              if (k < capacity) indices[k] = id;
              ++k;
              acc += (double)(id & 1023u);
            }
        start[0] += brick_size;
      }
      start[0] = bbox_min[0];
      start[1] += brick_size;
    }
    start[1] = bbox_min[1];
    start[2] += brick_size;
  }
  *count = k;
  return acc;
}

/* ------------------------------------------------------------------------- */
/* Calibration-volume file format (framework/calibration/calibration_volume.hpp:30-79) */

ORC_API int orc_lut_write(const char* path, const uint32_t* res, const float* limits,
                          const float* data, int floats_per_record)
{
  FILE* f = fopen(path, "wb");
  if (!f) return -1;
  size_t n = (size_t)res[0] * res[1] * res[2] * floats_per_record;
  int ok = fwrite(res, 4, 3, f) == 3 && fwrite(limits, 4, 2, f) == 2 && fwrite(data, 4, n, f) == n;
  fclose(f);
  return ok ? 0 : -2;
}

ORC_API int orc_lut_read_header(const char* path, uint32_t* res, float* limits)
{
  FILE* f = fopen(path, "rb");
  if (!f) return -1;
  int ok = fread(res, 4, 3, f) == 3 && fread(limits, 4, 2, f) == 2;
  fclose(f);
  return ok ? 0 : -2;
}

ORC_API int orc_lut_read(const char* path, float* data, int floats_per_record)
{
  FILE* f = fopen(path, "rb");
  if (!f) return -1;
  uint32_t res[3];
  float lim[2];
  int ok = fread(res, 4, 3, f) == 3 && fread(lim, 4, 2, f) == 2;
  size_t n = ok ? (size_t)res[0] * res[1] * res[2] * floats_per_record : 0;
  ok = ok && fread(data, 4, n, f) == n;
  fclose(f);
  return ok ? 0 : -2;
}
