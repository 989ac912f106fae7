// kernels_pre.hip -- depth-preprocessing chain for gfx950 (MI355X).
//
// One kernel per reference pass so per-pass parity stays checkable:
//   k_morph      glsl/pre_morph.fs       (NetKinectArray::processDepth, NetKinectArray.cpp:251-290)
//   k_pre_depth  glsl/pre_depth.fs + inc_color.glsl + inc_bbox_test.glsl   (:331-356)
//   k_boundary   glsl/pre_boundary.fs    (:362-377)
//   k_normal     glsl/pre_normal.fs + inc_bricks.glsl mark_brick           (:382-397)
//   k_quality    glsl/pre_quality.fs     (:400-414)
// The launch geometry replaces ScreenQuad (one fragment per depth texel,
// framework/rendering/screen_quad.cpp:7-35): 16x16 pixel blocks (4 wavefronts), the sensor layers interleaved
// along grid.x (block_pos).  The two 13x13 passes stage their depth
// window (16+12)^2 in LDS once per block instead of 169 texture fetches per pixel.
// These passes move ~56 B/pixel of compulsory traffic and are ALU/latency bound;
// they are reported as time, not as a roofline fraction.
#include <hip/hip_runtime.h>

#include "rgbdr_internal.hpp"
#include "dxt.cuh"
#include "sampling.cuh"

namespace rgbdr {

constexpr int BX = 16, BY = 16;   // pixel block
constexpr int R13 = 6;            // 13x13 window radius
constexpr int TW = BX + 2 * R13;  // 28
constexpr int TH = BY + 2 * R13;
// LDS row pitch of the 13x13 windows: a 32-lane ds_read_b32 group holds two 16-pixel rows of a wavefront;
// 48 = 16 (mod 32) puts them on disjoint halves of the 32 banks (pitch 29 spent 49 % of the LDS cycles on
// conflicts, profiles/r01_pmc_summary_v10.json)
constexpr int TPITCH = 48;

// Block -> (sensor layer, block column).  The sensors are interleaved along x (grid.x = N * columns) instead of
// being stacked in grid.z: the blocks that carry a surface -- and run the 169-tap loops, 29 of k_pre_depth's 53 us
// -- cluster in the same image region of every sensor, and with one sensor after the other they reached the
// machine in N bursts of about one heavy wavefront per SIMD, which then issues at a single wavefront's rate
// (dependencies and LDS waits exposed).  Interleaved, the heavy wavefronts of all sensors are resident together.
struct BlockPos {
  int l, bx, by;
};
__device__ __forceinline__ BlockPos block_pos(int N, int first = 0)
{
  BlockPos b;
  b.bx = (int)blockIdx.x / N;
  b.l = first + (int)blockIdx.x - b.bx * N;
  b.by = (int)blockIdx.y;
  return b;
}
// a launch covers the sensor layers [p.first, p.first + p.count): all of them, or this rank's shard of the pre_* chain
__device__ __forceinline__ BlockPos block_pos(const PreParams& p) { return block_pos(p.count, p.first); }
static dim3 pass_grid(const PreParams& p) { return dim3((unsigned)(((p.W + BX - 1) / BX) * p.count), (unsigned)((p.H + BY - 1) / BY), 1); }

#ifdef RGBDR_TRACE_BLOCKS
// Developer build (make EXTRA=-DRGBDR_TRACE_BLOCKS, profiles/pre_blocks_probe.py): where and when every wavefront of the two 13 x 13
// kernels ran.  Per wavefront {HW_ID, XCC_ID, s_memrealtime at entry (100 MHz), s_memrealtime at exit, shader clocks spent, flag}.
struct WaveTrace {
  uint32_t hw_id, xcc_id, t0, t1, clocks, flag;
};
__device__ WaveTrace g_wave_trace[2][4096 * 4];
struct TraceScope {
  int which;
  uint32_t t0, flag = 0;
  long long c0;
  __device__ TraceScope(int w) : which(w)
  {
    t0 = (uint32_t)wall_clock64();
    c0 = clock64();
  }
  __device__ ~TraceScope()
  {
    const int tid = threadIdx.y * BX + threadIdx.x;
    if (tid & 63) return;
    WaveTrace& t = g_wave_trace[which][((blockIdx.y * gridDim.x + blockIdx.x) * 4 + (tid >> 6)) % (4096 * 4)];
    t.hw_id = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
    t.xcc_id = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
    t.t0 = t0;
    t.t1 = (uint32_t)wall_clock64();
    t.clocks = (uint32_t)(clock64() - c0);
    t.flag = flag;
  }
};
extern "C" int rgbdr_debug_wave_trace(int which, void* dst, size_t bytes)
{
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_wave_trace), bytes, (size_t)which * sizeof(WaveTrace) * 4096 * 4, hipMemcpyDeviceToHost);
}
// experiment: launch position -> image block, per kernel (0xffff: natural order)
__device__ uint16_t g_block_order[2][4096];
__device__ int g_block_order_on[2];
extern "C" int rgbdr_debug_block_order(int which, const uint16_t* order, size_t n)
{
  int on = order != nullptr;
  if (order && hipMemcpyToSymbol(HIP_SYMBOL(g_block_order), order, n * 2, (size_t)which * 4096 * 2, hipMemcpyHostToDevice) != hipSuccess) return -1;
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_block_order_on), &on, 4, (size_t)which * 4, hipMemcpyHostToDevice);
}
__device__ __forceinline__ BlockPos block_pos_ordered(const PreParams& p, int which)
{
  int id = (int)(blockIdx.y * gridDim.x + blockIdx.x);
  if (g_block_order_on[which]) id = g_block_order[which][id];
  const int x = id % (int)gridDim.x;
  BlockPos b;
  b.bx = x / p.count;
  b.l = p.first + x - b.bx * p.count;
  b.by = id / (int)gridDim.x;
  return b;
}
#define BLOCK_POS(p, which) block_pos_ordered(p, which)
#define TRACE_SCOPE(which) TraceScope trace_(which)
#define TRACE_FLAG(v) trace_.flag = (v)
#else
#define BLOCK_POS(p, which) block_pos(p)
#define TRACE_SCOPE(which)
#define TRACE_FLAG(v)
#endif

// 1 - length(vec2(x,y)) * (1/6) for x,y in [-6,6], filled by the host with the
// same correctly-rounded sqrtf (pre_depth.fs:37-41,115)
__constant__ float c_gauss_space[169];

void set_gauss_table(const float* t) { (void)hipMemcpyToSymbol(HIP_SYMBOL(c_gauss_space), t, 169 * sizeof(float)); }

// ---------------------------------------------------------------------------
__global__ void k_u8_to_unit(const uint8_t* __restrict__ src, float* __restrict__ dst, size_t n)
{
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = (float)src[i] / 255.0f;  // GL_LUMINANCE u8 texture as the sampler returns it
}
void launch_u8_to_unit(const uint8_t* src, float* dst, size_t n, hipStream_t s)
{
  hipLaunchKernelGGL(k_u8_to_unit, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, dst, n);
}

__global__ void k_repack_xyz(const float* __restrict__ src, float4* __restrict__ dst, size_t n)
{
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = make_float4(src[3 * i], src[3 * i + 1], src[3 * i + 2], 0.0f);
}
void launch_repack_xyz(const float* src, float4* dst, size_t n, hipStream_t s)
{
  hipLaunchKernelGGL(k_repack_xyz, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, dst, n);
}

// ---------------------------------------------------------------------------
// DXT1 / DXT5 colour frames -> RGB8 at upload.  The GL driver decodes the
// compressed colour array in the reference (NetKinectArray.cpp:149-156); this
// follows the integer arithmetic of the reference's own CPU decode of the same
// frames, squish (NetKinectArray.cpp:633, external/squish/colourblock.cpp:140-214).
// One thread per 4x4 block.
__device__ __forceinline__ void unpack565(const uint8_t* p, int* c)
{
  const int value = (int)p[0] | ((int)p[1] << 8);
  const int r = (value >> 11) & 0x1f, g = (value >> 5) & 0x3f, b = value & 0x1f;
  c[0] = ((r << 3) | (r >> 2)) & 0xff;
  c[1] = ((g << 2) | (g >> 4)) & 0xff;
  c[2] = ((b << 3) | (b >> 2)) & 0xff;
}

// the four colours of a block (squish::DecompressColour, external/squish/colourblock.cpp:140-214)
__device__ __forceinline__ void dxt_palette(const uint8_t* src, int mode, int codes[4][3])
{
  unpack565(src, codes[0]);
  unpack565(src + 2, codes[1]);
  const int a = (int)src[0] | ((int)src[1] << 8), bb = (int)src[2] | ((int)src[3] << 8);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int c = codes[0][i], d = codes[1][i];
    if (mode == 1 && a <= bb) {
      codes[2][i] = (c + d) / 2;
      codes[3][i] = 0;
    } else {
      codes[2][i] = (2 * c + d) / 3;
      codes[3][i] = (c + 2 * d) / 3;
    }
  }
}

__global__ void k_decode_dxt(const uint8_t* __restrict__ blocks_all, int W, int H, int mode, size_t layer_bytes,
                             uint8_t* __restrict__ rgb_all)
{
  const int bw = (W + 3) / 4, bh = (H + 3) / 4;
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= bw * bh) return;
  const uint8_t* src = blocks_all + (size_t)blockIdx.y * layer_bytes + (size_t)b * (mode == 1 ? 8 : 16) + (mode == 1 ? 0 : 8);
  uint8_t* rgb = rgb_all + (size_t)blockIdx.y * W * H * 3;
  int codes[4][3];
  dxt_palette(src, mode, codes);
  const int x0 = (b % bw) * 4, y0 = (b / bw) * 4;
  // a block row is 12 bytes: three aligned words when the image rows are (W * 3 a multiple of 4; x0 * 3 is a
  // multiple of 12) and the block lies inside the image -- 12 word stores per block instead of 48 byte stores
  const bool words = ((W * 3) & 3) == 0 && x0 + 4 <= W && (((size_t)rgb) & 3) == 0;
  for (int py = 0; py < 4; ++py) {
    const int bits = src[4 + py];
    const int sy = y0 + py;
    if (sy >= H) continue;
    if (words) {
      uint32_t px[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int idx = (bits >> (2 * k)) & 3;
        px[k] = (uint32_t)codes[idx][0] | ((uint32_t)codes[idx][1] << 8) | ((uint32_t)codes[idx][2] << 16);
      }
      uint32_t* o = reinterpret_cast<uint32_t*>(rgb + ((size_t)sy * W + x0) * 3);
      o[0] = px[0] | (px[1] << 24);
      o[1] = (px[1] >> 8) | (px[2] << 16);
      o[2] = (px[2] >> 16) | (px[3] << 8);
      continue;
    }
    for (int px = 0; px < 4; ++px) {
      const int sx = x0 + px;
      if (sx >= W) continue;
      const int idx = (bits >> (2 * px)) & 3;
      uint8_t* o = rgb + ((size_t)sy * W + sx) * 3;
      o[0] = (uint8_t)codes[idx][0];
      o[1] = (uint8_t)codes[idx][1];
      o[2] = (uint8_t)codes[idx][2];
    }
  }
}
void launch_decode_dxt(const uint8_t* blocks, int W, int H, int mode, int N, size_t layer_bytes, uint8_t* rgb,
                       hipStream_t s)
{
  const int nb = ((W + 3) / 4) * ((H + 3) / 4);
  hipLaunchKernelGGL(k_decode_dxt, dim3((nb + 127) / 128, N), dim3(128), 0, s, blocks, W, H, mode, layer_bytes, rgb);
}

// ---------------------------------------------------------------------------
// pre_morph.fs mode 0 (dilate, :73-112).  The mode-1 pass of the reference is a
// plain copy (:130-131) of this result, so one kernel produces what
// m_textures_depth2.front holds after processDepth().
__device__ __forceinline__ bool morph_valid(float d) { return d > 0.5f && d < 4.5f; }

// pre_morph.fs mode 0 for one pixel of one layer
__device__ __forceinline__ float morph_pixel(const float* __restrict__ in, int W, int H, int px, int py)
{
  const float depth = in[(size_t)py * W + px];
  if (morph_valid(depth)) return depth;
  float nb[9];
#pragma unroll
  for (int y = -1; y < 2; ++y)
#pragma unroll
    for (int x = -1; x < 2; ++x)
      nb[(y + 1) * 3 + (x + 1)] = in[(size_t)clampi(py + y, 0, H - 1) * W + clampi(px + x, 0, W - 1)];
  float average = 0.0f, num = 0.0f;
  bool valid = false;
#pragma unroll
  for (int k = 0; k < 9; ++k)
    if (morph_valid(nb[k])) {
      valid = true;
      average += nb[k];
      num += 1.0f;
    }
  if (!valid) return 0.0f;
  average /= num;
  float nd = 0.0f;
  num = 0.0f;
  valid = false;
#pragma unroll
  for (int k = 0; k < 9; ++k)
    if (morph_valid(nb[k]) && fabsf(average - nb[k]) < 0.2f) {
      valid = true;
      nd += nb[k];
      num += 1.0f;
    }
  return valid ? nd / num : 0.0f;
}

// clearOccupiedBricks rides along in the first kernel of the chain (a separate fill launch costs more stream
// time than zeroing the 1 MiB): the counters are next touched by k_normal, launches later
__device__ __forceinline__ void zero_words(uint32_t* __restrict__ zero, unsigned nzero)
{
  if (!zero) return;
  const unsigned nthreads = gridDim.x * gridDim.y * gridDim.z * (BX * BY);
  const unsigned tid = ((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * (BX * BY) + threadIdx.y * BX + threadIdx.x;
  for (unsigned i = tid; i < nzero; i += nthreads) zero[i] = 0u;
}

__global__ __launch_bounds__(BX* BY) void k_morph(const float* __restrict__ in_all, float* __restrict__ out_all, int W,
                                                  int H, int first, int count, uint32_t* __restrict__ zero, unsigned nzero)
{
  zero_words(zero, nzero);
  const BlockPos bp = block_pos(count, first);
  const int px = bp.bx * BX + threadIdx.x, py = blockIdx.y * BY + threadIdx.y;
  if (px >= W || py >= H) return;
  const size_t lo = (size_t)bp.l * W * H;
  out_all[lo + (size_t)py * W + px] = morph_pixel(in_all + lo, W, H, px, py);
}

// A frame that is already on the device: the copy into the context's buffers and the morph pass in one launch (the
// 3x3 neighbourhood is read from the caller's buffer); `b_*`: the colour frame / DXT blocks, copied by the same lanes.
__global__ __launch_bounds__(BX* BY) void k_upload_morph(const float* __restrict__ src, float* __restrict__ raw,
                                                         float* __restrict__ morph, int W, int H, int first, int count,
                                                         const uint4* __restrict__ b_src, uint4* __restrict__ b_dst, size_t b_n16,
                                                         const uint8_t* b_tail_src, uint8_t* b_tail_dst, int b_tail)
{
  const size_t nthreads = (size_t)gridDim.x * gridDim.y * (BX * BY);
  const size_t tid = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (BX * BY) + threadIdx.y * BX + threadIdx.x;
  for (size_t i = tid; i < b_n16; i += nthreads) b_dst[i] = b_src[i];
  if (tid < (size_t)b_tail) b_tail_dst[tid] = b_tail_src[tid];
  const BlockPos bp = block_pos(count, first);
  const int px = bp.bx * BX + threadIdx.x, py = blockIdx.y * BY + threadIdx.y;
  if (px >= W || py >= H) return;
  const size_t lo = (size_t)bp.l * W * H, o = lo + (size_t)py * W + px;
  raw[o] = src[o];
  morph[o] = morph_pixel(src + lo, W, H, px, py);
}

void launch_morph(const PreParams& p, const float* in, float* out, uint32_t* zero, unsigned nzero, hipStream_t s)
{
  hipLaunchKernelGGL(k_morph, pass_grid(p), dim3(BX, BY), 0, s, in, out, p.W, p.H, p.first, p.count, zero, nzero);
}
// [first, first + count): the sensor layers whose raw depth is copied and morphed -- every sensor, or this rank's shard of the
// pre_* chain (the chain reads no other layer); the colour frame / DXT blocks of ALL sensors are copied either way (the
// slab ray-march shades from every sensor's colour)
bool launch_upload_morph(int W, int H, int first, int count, const void* depth_src, float* raw, float* morph, const void* b_src,
                         void* b_dst, size_t b_bytes, hipStream_t s)
{
  if ((((uintptr_t)depth_src & 3u) | (((uintptr_t)b_src | (uintptr_t)b_dst) & 15u)) != 0) return false;
  PreParams p{};
  p.W = W;
  p.H = H;
  p.N = p.count = count;
  const size_t b16 = b_bytes / 16;
  hipLaunchKernelGGL(k_upload_morph, pass_grid(p), dim3(BX, BY), 0, s, (const float*)depth_src, raw, morph, W, H, first, count,
                     (const uint4*)b_src, (uint4*)b_dst, b16, (const uint8_t*)b_src + b16 * 16, (uint8_t*)b_dst + b16 * 16,
                     (int)(b_bytes - b16 * 16));
  return true;
}

// ---------------------------------------------------------------------------
// x / C for a compile-time constant C as q = x * R, q + fma(-q, C, x) * R with R = RN(1 / C): three instructions
// instead of the ~12 of a correctly rounded division.  tests/const_division_check.c compares it with x / C for
// every binary32 x: for the constants used below the two differ only for |x| < 1e-30 (results near the
// denormal range), +-inf and -0.  Lab conversion never sees such an x: its input is a bilinear blend of u8
// colours / 255 with weights that are 0 or >= 2^-25 (the fraction of a coordinate near a texel centre is a
// multiple of half an ulp of 0.5), so every dividend below is +0, NaN or between 1e-22 and 1e3.
__device__ __forceinline__ float divc(float x, float C, float R)
{
  const float q = x * R;
  return __builtin_fmaf(__builtin_fmaf(-q, C, x), R, q);
}
#define RGBDR_DIVC(x, C) divc((x), (C), 1.0f / (C))

// inc_color.glsl.  With the reference's extra /255 (:14-16) a [0,1] colour never
// reaches either pow() branch; they are kept for inputs outside that range.
__device__ __forceinline__ float pivot_rgb(float n)
{
  return (n > 0.04045f ? powf((n + 0.055f) / 1.055f, 2.4f) : RGBDR_DIVC(n, 12.92f)) * 100.0f;
}
__device__ __forceinline__ float pivot_xyz(float n)
{
  return n > 0.008856f ? powf(n, 1.0f / 3.0f) : RGBDR_DIVC(903.3f * n + 16.0f, 116.0f);
}
__device__ __forceinline__ float3 rgb_to_lab(float3 rgb)
{
  const float r = pivot_rgb(RGBDR_DIVC(rgb.x, 255.0f)), g = pivot_rgb(RGBDR_DIVC(rgb.y, 255.0f)),
              b = pivot_rgb(RGBDR_DIVC(rgb.z, 255.0f));
  const float X = r * 0.4124f + g * 0.3576f + b * 0.1805f;
  const float Y = r * 0.2126f + g * 0.7152f + b * 0.0722f;
  const float Z = r * 0.0193f + g * 0.1192f + b * 0.9505f;
  const float x = pivot_xyz(RGBDR_DIVC(X, 95.047f)), y = pivot_xyz(RGBDR_DIVC(Y, 100.000f)), z = pivot_xyz(RGBDR_DIVC(Z, 108.883f));
  return make_float3(fmaxf(0.0f, 116.0f * y - 16.0f), 500.0f * (x - y), 200.0f * (y - z));
}

// `unorm` = LDS table of i / 255.0f (twelve correctly rounded divisions per pixel otherwise)
__device__ __forceinline__ float3 color_bilinear(const uint8_t* __restrict__ img, int W, int H, float u, float v,
                                                 const float* unorm)
{
  const Axis X = axis_linear(u, W), Y = axis_linear(v, H);
  const uint8_t* p00 = img + ((size_t)Y.i0 * W + X.i0) * 3;
  const uint8_t* p10 = img + ((size_t)Y.i0 * W + X.i1) * 3;
  const uint8_t* p01 = img + ((size_t)Y.i1 * W + X.i0) * 3;
  const uint8_t* p11 = img + ((size_t)Y.i1 * W + X.i1) * 3;
  float c[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float t00 = unorm[p00[k]], t10 = unorm[p10[k]];
    const float t01 = unorm[p01[k]], t11 = unorm[p11[k]];
    c[k] = lerpf(lerpf(t00, t10, X.a), lerpf(t01, t11, X.a), Y.a);
  }
  return make_float3(c[0], c[1], c[2]);
}

// the same lookup with the colour frame still in its DXT blocks: the four texels are decoded on the spot (what
// k_decode_dxt would have stored), so the 15 us decode launch and 16 MB of RGB8 per frame are only spent when a
// consumer asks for the decoded frame
__device__ __forceinline__ float3 color_bilinear_dxt(const uint8_t* __restrict__ layer, int W, int H, int mode, float u,
                                                     float v, const float* unorm)
{
  const Axis X = axis_linear(u, W), Y = axis_linear(v, H);
  const int bw = (W + 3) / 4;
  const int bx0 = X.i0 >> 2, bx1 = X.i1 >> 2, by0 = Y.i0 >> 2, by1 = Y.i1 >> 2;
  const uint2 b00 = dxt_block(layer, bw, mode, bx0, by0);
  const uint2 b10 = bx1 == bx0 ? b00 : dxt_block(layer, bw, mode, bx1, by0);
  const uint2 b01 = by1 == by0 ? b00 : dxt_block(layer, bw, mode, bx0, by1);
  const uint2 b11 = by1 == by0 ? b10 : (bx1 == bx0 ? b01 : dxt_block(layer, bw, mode, bx1, by1));
  int p00[3], p10[3], p01[3], p11[3];
  dxt_texel(b00, mode, X.i0, Y.i0, p00);
  dxt_texel(b10, mode, X.i1, Y.i0, p10);
  dxt_texel(b01, mode, X.i0, Y.i1, p01);
  dxt_texel(b11, mode, X.i1, Y.i1, p11);
  float c[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float t00 = unorm[p00[k]], t10 = unorm[p10[k]];
    const float t01 = unorm[p01[k]], t11 = unorm[p11[k]];
    c[k] = lerpf(lerpf(t00, t10, X.a), lerpf(t01, t11, X.a), Y.a);
  }
  return make_float3(c[0], c[1], c[2]);
}

// pre_depth.fs:51-72 sample(): optional u8 un-compress
__device__ __forceinline__ float pd_uncompress(float d, bool compress, float scale, float scaled_near, float near_)
{
  if (!compress) return d;
  if (d < scaled_near) return 0.0f;
  return (d * d + 0.15f * scaled_near) * scale + near_;
}

// Frame-independent lookups of pre_depth.fs for one sensor, evaluated once when its calibration is set:
// cc_far = texture(cv_uv, (u, v, 1.0)) -- what get_color's coordinate is for every depth_norm outside (0,1) --
// and whether texture(cv_xyz, (u, v, w)) lies in the box for a w that clamps to the first (bit 0) / last
// (bit 1) z plane.  Same expressions as the per-frame code, so the cached values are the per-frame values.
__global__ __launch_bounds__(BX* BY) void k_pre_cache(PreParams p, int l)
{
  const int px = blockIdx.x * BX + threadIdx.x, py = blockIdx.y * BY + threadIdx.y;
  const int W = p.W, H = p.H;
  if (px >= W || py >= H) return;
  const size_t o = (size_t)l * W * H + (size_t)py * W + px;
  const float u = ((float)px + 0.5f) / (float)W, v = ((float)py + 0.5f) / (float)H;
  p.cc_far[o] = tex3d_uv(p.cv_uv[l], p.uv_res[l][0], p.uv_res[l][1], p.uv_res[l][2], u, v, 1.0f);
  unsigned char flags = 0;
  for (int side = 0; side < 2; ++side) {
    const float w = side ? 2.0f : -1.0f;  // t = w * rz - 0.5 is below 0 / at least rz - 1 for every rz >= 1
    const float3 pw = tex3d_xyz(p.cv_xyz[l], p.xyz_res[l][0], p.xyz_res[l][1], p.xyz_res[l][2], 0, u, v, w);
    const bool in_box = pw.x >= p.bbox_min[0] && pw.y >= p.bbox_min[1] && pw.z >= p.bbox_min[2] &&
                        pw.x <= p.bbox_max[0] && pw.y <= p.bbox_max[1] && pw.z <= p.bbox_max[2];
    flags |= in_box ? (1u << side) : 0u;
  }
  p.box_flags[o] = flags;
}
void launch_pre_cache(const PreParams& p, int sensor, hipStream_t s)
{
  dim3 grid((p.W + BX - 1) / BX, (p.H + BY - 1) / BY, 1);
  hipLaunchKernelGGL(k_pre_cache, grid, dim3(BX, BY), 0, s, p, sensor);
}

__global__ __launch_bounds__(BX* BY) void k_pre_depth(PreParams p, uint32_t* __restrict__ zero, unsigned nzero)
{
  TRACE_SCOPE(0);
  zero_words(zero, nzero);  // (when the morph pass did not run in this chain: it was done with the upload)
  __shared__ float tile[TH][TPITCH];
  __shared__ float unorm[256];  // i / 255.0f: what the sampler returns for a u8 colour channel
  const BlockPos bp = BLOCK_POS(p, 0);
  const int l = bp.l;
  const int W = p.W, H = p.H;
  const float* depth = p.depth_in + (size_t)l * W * H;
  const bool compress = p.compress != 0;
  const float scale = p.far_[l] - p.near_[l];
  const float scaled_near = scale / 255.0f;
  const int bx0 = bp.bx * BX - R13, by0 = bp.by * BY - R13;
  // stage the (clamped) depth window once per block.  Taps outside [min_ds, max_ds] are skipped by
  // the filter (pre_depth.fs:100-103); they are staged as the finite sentinel 3e38 so that, for a centre
  // depth of ordinary magnitude, the range test |ds - depth| > dist_range_max alone rejects them (NaN taps
  // stay NaN, as they pass all three tests in the shader).  A non-finite or huge centre depth takes the
  // three-test loop.
  const float min_ds = p.cv_min_ds[l], max_ds = p.cv_max_ds[l];
  unorm[threadIdx.y * BX + threadIdx.x] = (float)(threadIdx.y * BX + threadIdx.x) / 255.0f;  // 256 threads, 256 entries
  const int px = bp.bx * BX + threadIdx.x, py = bp.by * BY + threadIdx.y;
  const bool inside = px < W && py < H;
  const size_t o = (size_t)l * W * H + (inside ? (size_t)py * W + px : 0);
  const float u = ((float)px + 0.5f) / (float)W, v = ((float)py + 0.5f) / (float)H;
  const float range = max_ds - min_ds;
  const float depth0 = pd_uncompress(inside ? depth[(size_t)py * W + px] : 0.0f, compress, scale, scaled_near, p.near_[l]);
  const float depth_norm = (depth0 - min_ds) / range;
  // A pixel without a measurement (depth 0, or anything nearer than the first / beyond the last z texel of
  // cv_xyz) looks cv_xyz up in its clamped first / last z plane, and every depth_norm outside (0,1) looks cv_uv
  // up at 1.0 (pre_depth.fs:136): both results depend on the pixel only, not on the frame, and were computed
  // with the same expressions when the calibration was set (k_pre_cache).  Two thirds of the benchmark pixels
  // take this path and skip both trilinear lookups (240 of the ~540 VALU instructions of this prologue).
  const int rz = p.xyz_res[l][2];
  const float tz = depth_norm * (float)rz - 0.5f;  // axis_linear's t on the z axis
  const unsigned char cached = inside ? p.box_flags[o] : 0;
  const bool first_plane = tz < 0.0f && tz > -3.0e38f, last_plane = tz >= (float)(rz - 1) && tz < 3.0e38f;
  // Only a pixel inside the box runs the filter, and for a pixel of the first / last plane that is known already:
  // a block without a candidate (two thirds of the blocks) does not stage the 28 x 28 window at all.
  const bool candidate = inside && p.filter && (first_plane ? (cached & 1u) != 0 : (last_plane ? (cached & 2u) != 0 : true));
  if (__syncthreads_or(candidate)) {
    for (int i = threadIdx.y * BX + threadIdx.x; i < TW * TH; i += BX * BY) {
      const int ty = i / TW, tx = i - ty * TW;
      const float d = depth[(size_t)clampi(by0 + ty, 0, H - 1) * W + clampi(bx0 + tx, 0, W - 1)];
      const float ds = pd_uncompress(d, compress, scale, scaled_near, p.near_[l]);
      tile[ty][tx] = ((ds < min_ds) || (ds > max_ds)) ? 3.0e38f : ds;
    }
    __syncthreads();
  }
  if (!inside) return;
  bool in_box;
  if (first_plane) {
    in_box = (cached & 1u) != 0;   // both z texels clamp to plane 0: lerp(c, c, a) = c for the finite a of a finite tz
  } else if (last_plane) {
    in_box = (cached & 2u) != 0;   // ... to plane rz - 1
  } else {
    const float3 pw = tex3d_xyz(p.cv_xyz[l], p.xyz_res[l][0], p.xyz_res[l][1], rz, 0, u, v, depth_norm);
    in_box = pw.x >= p.bbox_min[0] && pw.y >= p.bbox_min[1] && pw.z >= p.bbox_min[2] &&
             pw.x <= p.bbox_max[0] && pw.y <= p.bbox_max[1] && pw.z <= p.bbox_max[2];
  }
  TRACE_FLAG((uint32_t)__popcll(__ballot(in_box && p.filter)));  // lanes that run the tap loop
  float2 cc;
  if (depth_norm <= 0.0f || depth_norm >= 1.0f) {
    cc = p.cc_far[o];
  } else {
    cc = tex3d_uv(p.cv_uv[l], p.uv_res[l][0], p.uv_res[l][1], p.uv_res[l][2], u, v, depth_norm);
  }
  const float3 rgb = p.color_dxt ? color_bilinear_dxt(p.color_dxt + (size_t)l * p.color_layer_bytes, p.Wc, p.Hc, p.color_mode, cc.x, cc.y, unorm)
                                 : color_bilinear(p.color + (size_t)l * p.Wc * p.Hc * 3, p.Wc, p.Hc, cc.x, cc.y, unorm);
  const float3 lab = rgb_to_lab(rgb);
  p.lab[o * 3 + 0] = lab.x;
  p.lab[o * 3 + 1] = lab.y;
  p.lab[o * 3 + 2] = lab.z;
  float2 out;
  if (!in_box) {
    out = make_float2(0.0f, 0.0f);
  } else if (!p.filter) {
    out = make_float2(depth_norm, 1.0f);
  } else {
    // bilateral_filter, pre_depth.fs:85-127: same tap order (y outer, x inner)
    const float dist_range_max = 0.35f * (depth0 / 4.5f);
    const float dist_range_max_inv = 1.0f / dist_range_max;
    float depth_bf = 0.0f, w = 0.0f, w_range = 0.0f;
    if (fabsf(depth0) <= 1.0e20f) {
      // Branch-free form of the shader's `continue`: a rejected tap contributes gr = +0, i.e. w_range += 0,
      // w += gs * 0 (= +-0) and depth_bf += (+-0) * ds.  That leaves the three sums unchanged bit for bit: they
      // start at +0 and a float sum only becomes -0 from -0 + -0, and ds is finite for every rejected tap (the
      // sentinel, or a value inside the calibrated range) so (+-0) * ds is a zero.  No exec-mask juggling per tap.
      for (int y = 0; y < 13; ++y) {
#pragma unroll
        for (int x = 0; x < 13; ++x) {
          const float ds = tile[threadIdx.y + y][threadIdx.x + x];
          const float dr = fabsf(ds - depth0);
          float gr = 1.0f - fminf(dr, dist_range_max) * dist_range_max_inv;
          gr = (dr > dist_range_max) ? 0.0f : gr;  // also the staged sentinel of out-of-range taps
          const float ws = c_gauss_space[y * 13 + x] * gr;
          depth_bf += ws * ds;
          w += ws;
          w_range += gr;
        }
      }
    } else {  // inf / NaN / huge centre: the shader's three tests, on the window re-read from memory
      for (int y = 0; y < 13; ++y)
        for (int x = 0; x < 13; ++x) {
          const float d = depth[(size_t)clampi(py + y - R13, 0, H - 1) * W + clampi(px + x - R13, 0, W - 1)];
          const float ds = pd_uncompress(d, compress, scale, scaled_near, p.near_[l]);
          const float dr = fabsf(ds - depth0);
          if ((ds < min_ds) || (ds > max_ds) || (dr > dist_range_max)) continue;
          const float gs = c_gauss_space[y * 13 + x];
          const float gr = 1.0f - fminf(dr, dist_range_max) * dist_range_max_inv;
          const float ws = gs * gr;
          depth_bf += ws * ds;
          w += ws;
          w_range += gr;
        }
    }
    const float filtered = depth_bf / w;
    out = make_float2((filtered - min_ds) / range, w_range / 169.0f);
  }
  p.depth_rg[o * 2 + 0] = out.x;
  p.depth_rg[o * 2 + 1] = out.y;
}

void launch_pre_depth(const PreParams& p, uint32_t* zero, unsigned nzero, hipStream_t s)
{
  hipLaunchKernelGGL(k_pre_depth, pass_grid(p), dim3(BX, BY), 0, s, p, zero, nzero);
}

// ---------------------------------------------------------------------------
__device__ __forceinline__ float distance3(const float* a, const float* b)
{
  const float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
  return sqrtf(dx * dx + dy * dy + dz * dz);
}

// pre_boundary.fs.  Only pixels of uncertain range (dy <= 0.65: edge candidates, a few per cent) look at their
// 5x5 neighbourhood; a block that holds one stages the 20x20 window of depth_rg + Lab in LDS once (25 dependent
// global gathers per candidate before: 17 -> 7 us for the benchmark frames), the other blocks stage nothing.
__global__ __launch_bounds__(BX* BY) void k_boundary(PreParams p)
{
  constexpr int T = BX + 4;
  __shared__ float2 t_rg[T][T + 1];
  __shared__ float t_lab[T][T + 1][3];
  const BlockPos bp = block_pos(p);
  const int px = bp.bx * BX + threadIdx.x, py = blockIdx.y * BY + threadIdx.y;
  const int W = p.W, H = p.H;
  const size_t lo = (size_t)bp.l * W * H;
  const float* drg = p.depth_rg + lo * 2;
  const float* lab = p.lab + lo * 3;
  const bool inside = px < W && py < H;
  const size_t o = (size_t)(inside ? py : 0) * W + (inside ? px : 0);
  float dx = drg[o * 2], dy = drg[o * 2 + 1];
  const bool edge = inside && !(dx <= 0.0f) && !(dy > 0.65f);  // exactly the pixels that take the 5x5 branch below (NaN included)
  const bool any = __syncthreads_or(edge) != 0;
  if (any) {
    const int bx0 = bp.bx * BX - 2, by0 = blockIdx.y * BY - 2;
    for (int i = threadIdx.y * BX + threadIdx.x; i < T * T; i += BX * BY) {
      const int ty = i / T, tx = i - ty * T;
      const size_t os = (size_t)clampi(by0 + ty, 0, H - 1) * W + clampi(bx0 + tx, 0, W - 1);
      t_rg[ty][tx] = make_float2(drg[os * 2], drg[os * 2 + 1]);
      t_lab[ty][tx][0] = lab[os * 3];
      t_lab[ty][tx][1] = lab[os * 3 + 1];
      t_lab[ty][tx][2] = lab[os * 3 + 2];
    }
    __syncthreads();
  }
  if (!inside) return;
  float sil = 1.0f;
  if (dx <= 0.0f) {  // pre_boundary.fs:90-100
    dy = 0.0f;
    sil = 0.0f;
  } else if (!(dy > 0.65f)) {  // :102-113
    sil = 0.0f;
    const float* color = t_lab[threadIdx.y + 2][threadIdx.x + 2];
    float total = 0.0f, num = 0.0f;
    for (int y = 0; y < 5; ++y)
#pragma unroll
      for (int x = 0; x < 5; ++x) {
        const float2 s = t_rg[threadIdx.y + y][threadIdx.x + x];
        if (s.x > 0.0f && s.y > 0.65f) {
          num += 1.0f;
          total += distance3(color, t_lab[threadIdx.y + y][threadIdx.x + x]);
        }
      }
    const float color_dist = (num < 16.0f * 0.5f) ? 1.0f : total / num;
    if (color_dist > 0.5f || !p.refine) {
      dx = -1.0f;
      dy = 0.1f;
    } else {
      dy = 1.0f;
    }
  } else {
    dy = 0.0f;
  }
  p.depth_b_rg[(lo + o) * 2] = dx;
  p.depth_b_rg[(lo + o) * 2 + 1] = dy;
  p.silhouette[lo + o] = sil;
}

void launch_boundary(const PreParams& p, hipStream_t s)
{
  hipLaunchKernelGGL(k_boundary, pass_grid(p), dim3(BX, BY), 0, s, p);
}

// ---------------------------------------------------------------------------
__device__ __forceinline__ bool unit_outside(float d) { return (d <= 0.0f) || (d >= 1.0f); }
__device__ __forceinline__ float sign_of(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }

// Wave-aggregated counter increment: lanes of a wavefront that hit the same brick
// (a 16x4 pixel patch usually lands in one or two bricks) are combined into a single
// atomicAdd of their summed increments -- integer sums, so the counters are identical
// to per-pixel atomics.  `id` < 0 means "no increment from this lane".
__device__ __forceinline__ void wave_add(uint32_t* counters, int id, unsigned inc)
{
  unsigned long long todo = __ballot(id >= 0 && inc != 0u);
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const int lid = __shfl(id, leader);
    const unsigned long long same = __ballot(id == lid && inc != 0u) & todo;
    // sum of the increments of the matching lanes (increments are 0 or 1 here)
    if ((int)(threadIdx.y * BX + threadIdx.x) % 64 == leader) atomicAdd(&counters[lid], (unsigned)__popcll(same));
    todo &= ~same;
  }
}

// inc_bricks.glsl:40-58.  Positions whose home brick lies outside the brick grid
// are skipped (the reference indexes out of range there).  Returns the
// two counter ids (or -1) and the neighbour increment; the caller adds them wave-wide.
__device__ __forceinline__ void mark_brick(const PreParams& p, float3 pos, int& home_id, int& nb_id, unsigned& nb_inc)
{
  home_id = nb_id = -1;
  nb_inc = 0u;
  const float w[3] = {pos.x, pos.y, pos.z};
  int idx[3];
  float diff[3], dabs[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float f = floorf((w[a] - p.bbox_min[a]) / p.brick_size);
    if (!(f >= 0.0f) || !(f < (float)p.res_bricks[a])) return;
    idx[a] = (int)f;
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float center = (float)idx[a] * p.brick_size + p.bbox_min[a] + 0.5f * p.brick_size;
    diff[a] = w[a] - center;
    dabs[a] = fabsf(diff[a]);
  }
  const float min_v = fmaxf(dabs[0], fmaxf(dabs[1], dabs[2]));
  int nb[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float mc = (dabs[a] < min_v) ? 0.0f : 1.0f;
    nb[a] = clampi(idx[a] + (int)sign_of(diff[a] * mc), 0, p.res_bricks[a] - 1);
  }
  const int rx = p.res_bricks[0], ry = p.res_bricks[1];
  nb_inc = (dabs[0] > p.brick_size * 0.1f) ? 1u : 0u;
  nb_id = (nb[2] * ry + nb[1]) * rx + nb[0];
  home_id = (idx[2] * ry + idx[1]) * rx + idx[0];
}

// pre_normal.fs:26-56 for one pixel with a depth inside (0,1): `dt/dbm/dl/dr` are the depth_b values of the four
// neighbours (clamped at the image edge).  Returns the unit normal.
__device__ __forceinline__ float3 world_at(const PreParams& p, int l, int px, int py, float depth)
{
  const float u = ((float)px + 0.5f) / (float)p.W, v = ((float)py + 0.5f) / (float)p.H;
  return tex3d_xyz(p.cv_xyz[l], p.xyz_res[l][0], p.xyz_res[l][1], p.xyz_res[l][2], 0, u, v, depth);
}
template <bool PAIRS>
__device__ __forceinline__ float3 normal_at(const PreParams& p, int l, int px, int py, float depth, float dt, float dbm,
                                            float dl, float dr)
{
  const int W = p.W, H = p.H;
  const float u = ((float)px + 0.5f) / (float)W, v = ((float)py + 0.5f) / (float)H;
  const float tsx = 1.0f / (float)W, tsy = 1.0f / (float)H;
  const float4* lut = p.cv_xyz[l];
  const int rx = p.xyz_res[l][0], ry = p.xyz_res[l][1], rz = p.xyz_res[l][2];
  dt = unit_outside(dt) ? depth : dt;
  dbm = unit_outside(dbm) ? depth : dbm;
  dl = unit_outside(dl) ? depth : dl;
  dr = unit_outside(dr) ? depth : dr;
  // PAIRS: two rounds of two lookups, deliberately not unrolled, for the kernel that also runs the quality pass
  // (all five lookups in flight at once are 40 gathers = 108 VGPRs here and 178 there: two wavefronts per SIMD).
  // (Keeping the centre's eight LUT corners in registers for the neighbours, which mostly fall into the same
  // cell, measured slower: the repeated gathers hit L1 and the compare costs more.)
  float ax = 0.0f, ay = 0.0f, az = 0.0f, bx = 0.0f, by = 0.0f, bz = 0.0f;
  if (PAIRS) {
#pragma unroll 1
    for (int j = 0; j < 2; ++j) {
      const float3 w0 = tex3d_xyz(lut, rx, ry, rz, 0, j ? u + tsx : u, j ? v : v + tsy, j ? dr : dt);   // wt | wr
      const float3 w1 = tex3d_xyz(lut, rx, ry, rz, 0, j ? u - tsx : u, j ? v : v - tsy, j ? dl : dbm);  // wb | wl
      const float ex = w1.x - w0.x, ey = w1.y - w0.y, ez = w1.z - w0.z;
      if (j == 0) {
        ax = ex, ay = ey, az = ez;  // wb - wt
      } else {
        bx = ex, by = ey, bz = ez;  // wl - wr
      }
    }
  } else {
    const float3 wt = tex3d_xyz(lut, rx, ry, rz, 0, u, v + tsy, dt);
    const float3 wb = tex3d_xyz(lut, rx, ry, rz, 0, u, v - tsy, dbm);
    const float3 wl = tex3d_xyz(lut, rx, ry, rz, 0, u - tsx, v, dl);
    const float3 wr = tex3d_xyz(lut, rx, ry, rz, 0, u + tsx, v, dr);
    ax = wb.x - wt.x, ay = wb.y - wt.y, az = wb.z - wt.z;
    bx = wl.x - wr.x, by = wl.y - wr.y, bz = wl.z - wr.z;
  }
  const float cx = ay * bz - by * az, cy = az * bx - bz * ax, cz = ax * by - bx * ay;
  const float len = sqrtf(cx * cx + cy * cy + cz * cz);
  return make_float3(cx / len, cy / len, cz / len);
}

__global__ __launch_bounds__(BX* BY) void k_normal(PreParams p)
{
  const BlockPos bp = block_pos(p);
  const int px = bp.bx * BX + threadIdx.x, py = blockIdx.y * BY + threadIdx.y;
  const int W = p.W, H = p.H, l = bp.l;
  if (px >= W || py >= H) return;
  const size_t lo = (size_t)l * W * H;
  const float* db = p.depth_b_rg + lo * 2;
  const size_t o = (size_t)py * W + px;
  float3 n = make_float3(0.0f, 0.0f, 0.0f);
  const float depth = db[o * 2];
  int home_id = -1, nb_id = -1;
  unsigned nb_inc = 0u;
  if (!unit_outside(depth)) {
    const float dt = db[((size_t)clampi(py + 1, 0, H - 1) * W + px) * 2];
    const float dbm = db[((size_t)clampi(py - 1, 0, H - 1) * W + px) * 2];
    const float dl = db[((size_t)py * W + clampi(px - 1, 0, W - 1)) * 2];
    const float dr = db[((size_t)py * W + clampi(px + 1, 0, W - 1)) * 2];
    if (p.brick_counters) mark_brick(p, world_at(p, l, px, py, depth), home_id, nb_id, nb_inc);
    n = normal_at<false>(p, l, px, py, depth, dt, dbm, dl, dr);
  }
  p.normal[(lo + o) * 3 + 0] = n.x;
  p.normal[(lo + o) * 3 + 1] = n.y;
  p.normal[(lo + o) * 3 + 2] = n.z;
  if (p.brick_counters) {  // reconverged: every live lane of the wavefront takes part
    wave_add(p.brick_counters, nb_id, nb_inc);
    wave_add(p.brick_counters, home_id, 1u);
  }
}

void launch_normal(const PreParams& p, hipStream_t s)
{
  hipLaunchKernelGGL(k_normal, pass_grid(p), dim3(BX, BY), 0, s, p);
}

// ---------------------------------------------------------------------------
// pre_quality.fs.  The 28x28 depth_b window of a block in LDS: taps outside (0,1) count as border
// (pre_quality.fs:62-66) and are staged as +inf: for a centre depth inside (0,1) -- the only pixels that run the
// loop -- |inf - depth| = inf exceeds the range limit, so the one range test classifies them without the two
// bound tests.  NaN stays NaN (it is not "outside" in the shader either).
__device__ __forceinline__ void stage_depth_b(float (*tile)[TPITCH], const float* __restrict__ db, int bx0, int by0, int W,
                                              int H)
{
  for (int i = threadIdx.y * BX + threadIdx.x; i < TW * TH; i += BX * BY) {
    const int ty = i / TW, tx = i - ty * TW;
    const float d = db[((size_t)clampi(by0 + ty, 0, H - 1) * W + clampi(bx0 + tx, 0, W - 1)) * 2];
    tile[ty][tx] = unit_outside(d) ? __builtin_inff() : d;
  }
}

// lateral^6 * range^6 / (6.5 depth) of pre_quality.fs:64-112 for a centre depth inside (0,1).
// Branch-free: a border tap (outside (0,1): staged +inf, or beyond the range threshold) adds 1 to an integer count
// (exact up to 169) and +0 to w_range, which never changes a float sum that started at +0.
__device__ __forceinline__ float quality_taps(const float (*tile)[TPITCH], float depth)
{
  const float dist_range_max = 0.35f * (depth / 1.0f);
  const float dist_range_max_inv = 1.0f / dist_range_max;
  float w_range = 0.0f;
  int nborder = 0;
  for (int y = 0; y < 13; ++y) {
#pragma unroll
    for (int x = 0; x < 13; ++x) {
      const float ds = tile[threadIdx.y + y][threadIdx.x + x];
      const float dr = fabsf(ds - depth);
      const bool is_border = dr > dist_range_max;
      const float gr = 1.0f - fminf(dr, dist_range_max) * dist_range_max_inv;
      nborder += is_border ? 1 : 0;
      w_range += is_border ? 0.0f : gr;
    }
  }
  const float border = (float)nborder;
  const float lateral = 1.0f - border / 169.0f;
  const float l2 = lateral * lateral, l4 = l2 * l2;
  float q = l4 * l2;
  const float wr = w_range / 169.0f;
  const float w2 = wr * wr, w4 = w2 * w2;
  q *= w4 * w2;
  q /= depth * 6.5f;
  return q;
}

// pre_quality.fs:114-118: squared cosine between the view ray of the pixel's position and its normal
__device__ __forceinline__ float quality_angle(const PreParams& p, int l, float3 wp, float3 nrm)
{
  const float dx = p.cam_pos[l][0] - wp.x, dy = p.cam_pos[l][1] - wp.y, dz = p.cam_pos[l][2] - wp.z;
  const float len = sqrtf(dx * dx + dy * dy + dz * dz);
  const float angle = (dx / len) * nrm.x + (dy / len) * nrm.y + (dz / len) * nrm.z;
  return angle * angle;
}

// quality image + the packed 8-B texel the integration kernel samples: depth_b.r and the quality
// with "silhouette == 0" folded into its sign bit.  quality is a product of
// non-negative factors (or NaN), so the sign bit is free; |NaN| stays NaN.
__device__ __forceinline__ void store_quality_sil(const PreParams& p, size_t i, float depth, float q, float sil)
{
  p.quality[i] = q;
  const unsigned qbits = (__float_as_uint(q) & 0x7fffffffu) | (sil < 1.0f ? 0x80000000u : 0u);
  p.frame[i] = make_uint2(__float_as_uint(depth), qbits);
}
__device__ __forceinline__ void store_quality(const PreParams& p, size_t i, float depth, float q)
{
  store_quality_sil(p, i, depth, q, p.silhouette[i]);
}

__global__ __launch_bounds__(BX* BY) void k_quality(PreParams p)
{
  __shared__ float tile[TH][TPITCH];
  const BlockPos bp = block_pos(p);
  const int l = bp.l;
  const int W = p.W, H = p.H;
  const size_t lo = (size_t)l * W * H;
  const float* db = p.depth_b_rg + lo * 2;
  stage_depth_b(tile, db, bp.bx * BX - R13, blockIdx.y * BY - R13, W, H);
  __syncthreads();
  const int px = bp.bx * BX + threadIdx.x, py = blockIdx.y * BY + threadIdx.y;
  if (px >= W || py >= H) return;
  const size_t o = (size_t)py * W + px;
  const float depth = db[o * 2];
  float q = 0.0f;
  if (!unit_outside(depth)) {
    q = quality_taps(tile, depth);
    const float3 wp = world_at(p, l, px, py, depth);
    const float* nrm = p.normal + (lo + o) * 3;
    q *= quality_angle(p, l, wp, make_float3(nrm[0], nrm[1], nrm[2]));
  }
  store_quality(p, lo + o, depth, q);
}

void launch_quality(const PreParams& p, hipStream_t s)
{
  hipLaunchKernelGGL(k_quality, pass_grid(p), dim3(BX, BY), 0, s, p);
}

// ---------------------------------------------------------------------------
// pre_normal.fs + pre_quality.fs in one launch (the default; the two kernels above run when a host asks for the
// per-pass timers).  quality needs the normal of its own pixel only, and both need the pixel's position at its
// depth_b: one launch, one cv_xyz lookup and no normal re-read less, and the wavefronts waiting on the five
// lookups of the normal share their SIMD with wavefronts in the 169-tap loop.  Same expressions, same images.
template <int WAVES>
__global__ __launch_bounds__(BX* BY, WAVES) void k_normal_quality(PreParams p)
{
  __shared__ float tile[TH][TPITCH];
  const BlockPos bp = block_pos(p);
  const int l = bp.l;
  const int W = p.W, H = p.H;
  const size_t lo = (size_t)l * W * H;
  const float* db = p.depth_b_rg + lo * 2;
  const int px = bp.bx * BX + threadIdx.x, py = blockIdx.y * BY + threadIdx.y;
  const bool inside = px < W && py < H;
  const size_t o = inside ? (size_t)py * W + px : 0;
  const float depth = inside ? db[o * 2] : 0.0f;
  // two thirds of the blocks hold no pixel with a depth in (0,1) -- nothing to compute, nothing to stage: they
  // write their zeros and leave (the window is 784 texels for 256 pixels)
  if (!__syncthreads_or(inside && !unit_outside(depth))) {
    if (!inside) return;
    p.normal[(lo + o) * 3 + 0] = 0.0f;
    p.normal[(lo + o) * 3 + 1] = 0.0f;
    p.normal[(lo + o) * 3 + 2] = 0.0f;
    store_quality(p, lo + o, depth, 0.0f);
    return;
  }
  stage_depth_b(tile, db, bp.bx * BX - R13, blockIdx.y * BY - R13, W, H);
  __syncthreads();
  if (!inside) return;
  float3 n = make_float3(0.0f, 0.0f, 0.0f);
  float q = 0.0f;
  int home_id = -1, nb_id = -1;
  unsigned nb_inc = 0u;
  if (!unit_outside(depth)) {
    // the staged window holds the (edge-clamped) neighbours; +inf there means "outside (0,1)", which is
    // what normal_at tests for
    const int cy = threadIdx.y + R13, cx = threadIdx.x + R13;
    const float3 world = world_at(p, l, px, py, depth);
    if (p.brick_counters) mark_brick(p, world, home_id, nb_id, nb_inc);
    n = normal_at<true>(p, l, px, py, depth, tile[cy + 1][cx], tile[cy - 1][cx], tile[cy][cx - 1], tile[cy][cx + 1]);
    q = quality_taps(tile, depth);
    q *= quality_angle(p, l, world, n);
  }
  p.normal[(lo + o) * 3 + 0] = n.x;
  p.normal[(lo + o) * 3 + 1] = n.y;
  p.normal[(lo + o) * 3 + 2] = n.z;
  store_quality(p, lo + o, depth, q);
  if (p.brick_counters) {  // reconverged: every live lane of the wavefront takes part
    wave_add(p.brick_counters, nb_id, nb_inc);
    wave_add(p.brick_counters, home_id, 1u);
  }
}

// ---------------------------------------------------------------------------
// pre_boundary.fs + pre_normal.fs + pre_quality.fs in one launch (the default of process_textures; the separate
// kernels above run when a host asks for the per-pass timers).  The boundary pass only changes the depth of a
// pixel that is an edge candidate (depth > 0 and range weight <= 0.65: a few per cent of the pixels), from a 5 x 5
// neighbourhood of (depth, weight, Lab); everything else is a function of the pixel itself.  So the block stages
// the (depth, weight) window of its 28 x 28 depth_b window plus the 2-texel rim those neighbourhoods reach, the Lab
// window only when the depth_b window holds a candidate, and derives the depth_b window from them -- each value is
// what k_boundary writes at the clamped image position, computed with the same expressions in the same order.
// Its own 16 x 16 pixels' depth_b / silhouette are written from the same function.  One launch and one round trip
// of depth_b through memory less per frame.
constexpr int BR = R13 + 2;        // rim of the staged (depth, weight) / Lab windows
constexpr int BW = BX + 2 * BR;    // 32
constexpr int BP = BW + 1;         // LDS pitch in texels

// pre_boundary.fs:90-113 for the texel at LDS position (cy, cx)
__device__ __forceinline__ void boundary_texel(const float2 (*rg)[BP], const float (*lab)[BP][3], int cy, int cx, bool refine,
                                               float& dx, float& dy, float& sil)
{
  const float2 c = rg[cy][cx];
  dx = c.x;
  dy = c.y;
  sil = 1.0f;
  if (dx <= 0.0f) {
    dy = 0.0f;
    sil = 0.0f;
  } else if (!(dy > 0.65f)) {
    sil = 0.0f;
    const float* color = lab[cy][cx];
    float total = 0.0f, num = 0.0f;
    for (int y = 0; y < 5; ++y)
#pragma unroll
      for (int x = 0; x < 5; ++x) {
        const float2 t = rg[cy - 2 + y][cx - 2 + x];
        if (t.x > 0.0f && t.y > 0.65f) {
          num += 1.0f;
          total += distance3(color, lab[cy - 2 + y][cx - 2 + x]);
        }
      }
    const float color_dist = (num < 16.0f * 0.5f) ? 1.0f : total / num;
    if (color_dist > 0.5f || !refine) {
      dx = -1.0f;
      dy = 0.1f;
    } else {
      dy = 1.0f;
    }
  } else {
    dy = 0.0f;
  }
}

template <int WAVES>
__global__ __launch_bounds__(BX* BY, WAVES) void k_boundary_normal_quality(PreParams p)
{
  TRACE_SCOPE(1);
  __shared__ float tile[TH][TPITCH];
  __shared__ float2 w_rg[BW][BP];
  __shared__ float w_lab[BW][BP][3];
  const BlockPos bp = BLOCK_POS(p, 1);
  const int l = bp.l;
  const int W = p.W, H = p.H;
  const size_t lo = (size_t)l * W * H;
  const float* drg = p.depth_rg + lo * 2;
  const float* lab = p.lab + lo * 3;
  const int px = bp.bx * BX + threadIdx.x, py = bp.by * BY + threadIdx.y;
  const bool inside = px < W && py < H;
  const size_t o = inside ? (size_t)py * W + px : 0;
  const float2* drg2 = reinterpret_cast<const float2*>(drg);  // (depth, weight) pairs: 8-byte aligned (hipMalloc + a whole layer)
  const float2 own = inside ? drg2[o] : make_float2(0.0f, 1.0f);
  const float dx0 = own.x, dy0 = own.y;
  const bool cand0 = inside && !(dx0 <= 0.0f) && !(dy0 > 0.65f);
  // no pixel of the block is an edge candidate or keeps a depth in (0,1): depth_b = (depth, 0), the silhouette follows
  // the sign of the depth, normals and quality are zero -- nothing to stage
  if (!__syncthreads_or(inside && (cand0 || !unit_outside(dx0)))) {
    if (!inside) return;
    const float sil = dx0 <= 0.0f ? 0.0f : 1.0f;
    p.depth_b_rg[(lo + o) * 2] = dx0;
    p.depth_b_rg[(lo + o) * 2 + 1] = 0.0f;
    p.silhouette[lo + o] = sil;
    p.normal[(lo + o) * 3 + 0] = 0.0f;
    p.normal[(lo + o) * 3 + 1] = 0.0f;
    p.normal[(lo + o) * 3 + 2] = 0.0f;
    store_quality_sil(p, lo + o, dx0, 0.0f, sil);
    return;
  }
  const int bx0 = bp.bx * BX - BR, by0 = bp.by * BY - BR;
  const int tid = threadIdx.y * BX + threadIdx.x;
  bool cand = false;
  for (int i = tid; i < BW * BW; i += BX * BY) {
    const int ty = i / BW, tx = i - ty * BW;
    const size_t os = (size_t)clampi(by0 + ty, 0, H - 1) * W + clampi(bx0 + tx, 0, W - 1);
    const float2 v = drg2[os];
    w_rg[ty][tx] = v;
    cand |= ty >= 2 && ty < BW - 2 && tx >= 2 && tx < BW - 2 && !(v.x <= 0.0f) && !(v.y > 0.65f);
  }
  if (__syncthreads_or(cand)) {  // (also the barrier behind the staging above)
    for (int i = tid; i < BW * BW; i += BX * BY) {
      const int ty = i / BW, tx = i - ty * BW;
      const size_t os = (size_t)clampi(by0 + ty, 0, H - 1) * W + clampi(bx0 + tx, 0, W - 1);
      w_lab[ty][tx][0] = lab[os * 3];
      w_lab[ty][tx][1] = lab[os * 3 + 1];
      w_lab[ty][tx][2] = lab[os * 3 + 2];
    }
    __syncthreads();
  }
  // depth_b.r over the 28 x 28 window: the value at the CLAMPED image position (what stage_depth_b reads), whose
  // 5 x 5 neighbourhood is staged around it (positions beyond the image repeat the edge texels, as k_boundary stages them)
  for (int i = tid; i < TW * TH; i += BX * BY) {
    const int ty = i / TW, tx = i - ty * TW;
    const int cy = clampi(by0 + 2 + ty, 0, H - 1) - by0, cx = clampi(bx0 + 2 + tx, 0, W - 1) - bx0;
    float dx, dy, sil;
    boundary_texel(w_rg, w_lab, cy, cx, p.refine != 0, dx, dy, sil);
    tile[ty][tx] = unit_outside(dx) ? __builtin_inff() : dx;
  }
  __syncthreads();
  if (!inside) return;
  float depth, dyb, sil;
  boundary_texel(w_rg, w_lab, threadIdx.y + BR, threadIdx.x + BR, p.refine != 0, depth, dyb, sil);
  p.depth_b_rg[(lo + o) * 2] = depth;
  p.depth_b_rg[(lo + o) * 2 + 1] = dyb;
  p.silhouette[lo + o] = sil;
  float3 n = make_float3(0.0f, 0.0f, 0.0f);
  float q = 0.0f;
  int home_id = -1, nb_id = -1;
  unsigned nb_inc = 0u;
  if (!unit_outside(depth)) {
    const int cy = threadIdx.y + R13, cx = threadIdx.x + R13;
    const float3 world = world_at(p, l, px, py, depth);
    if (p.brick_counters) mark_brick(p, world, home_id, nb_id, nb_inc);
    n = normal_at<true>(p, l, px, py, depth, tile[cy + 1][cx], tile[cy - 1][cx], tile[cy][cx - 1], tile[cy][cx + 1]);
    q = quality_taps(tile, depth);
    q *= quality_angle(p, l, world, n);
  }
  p.normal[(lo + o) * 3 + 0] = n.x;
  p.normal[(lo + o) * 3 + 1] = n.y;
  p.normal[(lo + o) * 3 + 2] = n.z;
  store_quality_sil(p, lo + o, depth, q, sil);
  if (p.brick_counters) {  // reconverged: every live lane of the wavefront takes part
    wave_add(p.brick_counters, nb_id, nb_inc);
    wave_add(p.brick_counters, home_id, 1u);
  }
}

void launch_boundary_normal_quality(const PreParams& p, hipStream_t s)
{
  hipLaunchKernelGGL(k_boundary_normal_quality<4>, pass_grid(p), dim3(BX, BY), 0, s, p);
}

void launch_normal_quality(const PreParams& p, hipStream_t s)
{
  // four wavefronts per SIMD (114 VGPRs); three and five measured the same, two and six slower
  hipLaunchKernelGGL(k_normal_quality<4>, pass_grid(p), dim3(BX, BY), 0, s, p);
}

// ---------------------------------------------------------------------------
// updateOccupiedBricks (recon_integration.cpp:431-446) without the readback:
// mask[i] = counter[i] >= min_voxels.  The occupied count / ratio / id list are only
// produced when a consumer asks (k_compact_occupied), not every frame.
__global__ void k_update_occupied(const uint32_t* __restrict__ counters, uint32_t n, uint32_t min_voxels,
                                  uint8_t* __restrict__ mask)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) mask[i] = counters[i] >= min_voxels ? 1 : 0;
}
void launch_update_occupied(const uint32_t* counters, uint32_t n, uint32_t min_voxels, uint8_t* mask, uint32_t* count,
                            hipStream_t s)
{
  (void)count;
  hipLaunchKernelGGL(k_update_occupied, dim3((n + 255) / 256), dim3(256), 0, s, counters, n, min_voxels, mask);
}

// ascending id list for consumers (m_bricks_occupied): one block, chunked scan
__global__ __launch_bounds__(1024) void k_compact_occupied(const uint8_t* __restrict__ mask, uint32_t n,
                                                           uint32_t* __restrict__ ids, uint32_t* __restrict__ count)
{
  __shared__ uint32_t wave_sums[16];
  __shared__ uint32_t base;
  if (threadIdx.x == 0) base = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (uint32_t start = 0; start < n; start += 1024) {
    const uint32_t i = start + threadIdx.x;
    const bool occ = i < n && mask[i];
    const unsigned long long b = __ballot(occ);
    const uint32_t before = (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
    if (lane == 0) wave_sums[wave] = (uint32_t)__popcll(b);
    __syncthreads();
    uint32_t off = base;
    for (int w = 0; w < wave; ++w) off += wave_sums[w];
    if (occ) ids[off + before] = i;
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t t = 0;
      for (int w = 0; w < 16; ++w) t += wave_sums[w];
      base += t;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) *count = base;
}
void launch_compact_occupied(const uint8_t* mask, uint32_t n, uint32_t* ids, uint32_t* count, hipStream_t s)
{
  hipLaunchKernelGGL(k_compact_occupied, dim3(1), dim3(1024), 0, s, mask, n, ids, count);
}

}  // namespace rgbdr
