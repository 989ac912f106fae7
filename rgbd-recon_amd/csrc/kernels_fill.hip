// kernels_fill.hip -- hole filling of the ray-marched frame:
// ReconIntegration::fillColors (framework/reconstruction/recon_integration.cpp:280-339)
// with glsl/framebuffer_transfer.fs, glsl/tsdf_inpaint.fs, glsl/tsdf_colorfill.fs over
// the LOD atlas of ViewLod (framework/rendering/view_lod.cpp:24-61).
//
// The reference ping-pongs two atlases of 1.5*W x H texels: "native" (N) accumulates the
// LODs, "squeezed" (S) is the copy framebuffer_transfer.fs redraws after every LOD
// (S[x, y] = N[(int)((x + .5) / W * FW), (int)((y + .5) / H * H)] for x < W, the clear
// colour beyond), and tsdf_inpaint.fs reads S.  Here neither atlas is materialised:
//   * S is a pure index map of N, so the inpaint taps go through the map (same float
//     expressions, same texels);
//   * N's LOD-0 viewport is the ray-marched frame itself, read where it lies; only the
//     column band x >= W (the LODs >= 1, "atlasR", FW - W texels wide) is stored;
//   * during pass i the state of N the shader sees is: LODs < i computed, everything else
//     still the clear colour -- a tap that lands in LOD i's own viewport (the one being
//     written) or under it returns the clear colour, whatever the band's memory holds;
//   * where the 16 taps of a texel lie depends on the viewport alone: the N column of the 4
//     tap columns per texel column and the N row of the 4 tap rows per texel row are tables
//     (make_fill_tables, the shader's float expressions evaluated once per viewport size).
//     A wavefront issues one VALU instruction per 4 cycles, so a pass with few texels lasts
//     as long as one texel's instruction stream: 3000 instructions were 7 us per LOD.
// Launches per frame: LOD 1 (+ clear of the band below it), one per LOD whose viewport is
// larger than the tail's first, ONE workgroup for the tail (its LODs live in LDS between
// __syncthreads(), no global-memory phase), colorfill.  1280 x 720: 21 -> 6 launches.
#include <hip/hip_runtime.h>

#include "fill_taps.cuh"
#include "rgbdr_internal.hpp"
#include "sampling.cuh"

namespace rgbdr {

// The native atlas as the shaders see it, without its LOD-0 viewport being a copy: columns < W are the frame,
// columns >= W the stored band (AW = FW - W texels wide).
struct FillSrc {
  const float4* __restrict__ fcol;  // W x H ray-marched frame
  const float* __restrict__ fdep;
  const float4* __restrict__ acol;  // AW x H band holding the LODs >= 1
  const float* __restrict__ adep;
};

__device__ __forceinline__ float4 fc_clear_col() { return make_float4(0.0f, 1.0f, 0.0f, 0.0f); }  // glClearColor(0,1,0,0)

// texelFetch(N, (x, y)) -- outside the texture: 0
__device__ __forceinline__ void fc_fetch(const FillLayout& L, const FillSrc& S, int x, int y, float4& c, float& d)
{
  if (x < 0 || y < 0 || x >= L.FW || y >= L.H) {
    c = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    d = 0.0f;
  } else if (x < L.W) {
    c = S.fcol[(size_t)y * L.W + x];
    d = S.fdep[(size_t)y * L.W + x];
  } else {
    const int AW = L.FW - L.W;
    c = S.acol[(size_t)y * AW + (x - L.W)];
    d = S.adep[(size_t)y * AW + (x - L.W)];
  }
}

// The tail workgroup's copy of the part of N its passes read: columns [x0, x0 + w) x rows [y0, y0 + h) -- the band's rows
// from the last LOD's first to the first LOD's last, as wide as the first LOD, plus a rim of 3 texels to the left (frame
// columns), to the right and above (the LOD before the first), which is as far as a 4 x 4 neighbourhood reaches.
struct FillLds {
  const float4* col;
  const float* dep;
  int x0, y0, w, h;
};

// tsdf_inpaint.fs for one texel of LOD i, given the N columns xe[4] / rows ye[4] of its taps (fill_taps.cuh), reading the
// native atlas in its state before this pass: 32 loads in flight, then the shader's two sums in its order.  A tap outside
// the atlas or on the clear colour has alpha 0 -- it takes part in nothing but the depth of the num_samples == 0 branch
// (centre tap), so its loaded value is dropped, not replaced.  LDS: taps inside M come from there, and the global loads
// are issued only if some lane of the wavefront has a tap elsewhere.
template <bool LDS>
__device__ __forceinline__ void fc_inpaint_texel(const FillLayout& L, const FillSrc& S, const int (&xe)[4], const int (&ye)[4],
                                                 const FillLds& M, float4& oc, float& od)
{
  const int AW = L.FW - L.W;
  bool xz[4], xc[4], xb[4], xl[4], yz[4], yc[4], yl[4];
  int colv[4], rowf[4], rowb[4], coll[4], rowl[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    xz[t] = xe[t] == FC_TAP_OUTSIDE;
    xc[t] = xe[t] == FC_TAP_CLEAR;
    xb[t] = xe[t] >= L.W;
    colv[t] = xe[t] < 0 ? 0 : (xb[t] ? xe[t] - L.W : xe[t]);
    xl[t] = LDS && xe[t] >= M.x0 && xe[t] < M.x0 + M.w;
    coll[t] = xe[t] - M.x0;
    yz[t] = ye[t] < 0;
    yc[t] = !yz[t] && (ye[t] & FC_ROW_CLEAR) != 0;
    const int ny = yz[t] ? 0 : (ye[t] & ~FC_ROW_CLEAR);
    rowf[t] = ny * L.W;
    rowb[t] = ny * AW;
    yl[t] = LDS && !yz[t] && ny >= M.y0 && ny < M.y0 + M.h;
    rowl[t] = (ny - M.y0) * M.w;
  }
  float4 cs[16];
  float ds[16];
  bool live[16], inl[16];
  bool need_global = !LDS;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int x = k & 3, y = k >> 2;
    const bool zero = xz[x] || yz[y];
    const bool clear = xc[x] || (xb[x] && yc[y]);
    live[k] = !zero && !clear;
    inl[k] = live[k] && xl[x] && yl[y];
    if (LDS) need_global = need_global || (live[k] && !inl[k]);
  }
  if (!LDS || __builtin_amdgcn_ballot_w64(need_global) != 0) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int x = k & 3, y = k >> 2;
      const unsigned at = (unsigned)((xb[x] ? rowb[y] : rowf[y]) + colv[x]);  // a texel of the frame / band whatever the kind
      cs[k] = (xb[x] ? S.acol : S.fcol)[at];
      ds[k] = (xb[x] ? S.adep : S.fdep)[at];
    }
  } else {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      cs[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      ds[k] = 0.0f;
    }
  }
  if (LDS) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int x = k & 3, y = k >> 2;
      const int lt = inl[k] ? rowl[y] + coll[x] : 0;
      const float4 lc = M.col[lt];
      const float ld = M.dep[lt];
      if (inl[k]) {
        cs[k] = lc;
        ds[k] = ld;
      }
    }
  }
  // a sum that starts at +0 is never -0, so adding +0 for a tap that does not count leaves the shader's sum bit for bit
  float depth_av = 0.0f;
  int num = 0;
#pragma unroll
  for (int x = 0; x < 4; ++x)  // the shader's order of accumulation: x outer, y inner
#pragma unroll
    for (int y = 0; y < 4; ++y) {
      const int k = x + y * 4;
      live[k] = live[k] && !(cs[k].w <= 0.0f);
      depth_av += live[k] ? ds[k] : 0.0f;
      num += live[k] ? 1 : 0;
    }
  if (num == 0) {
    const int k = 1 + 1 * 4;  // texelFetch(texture_depth, pos_int)
    const bool zero = xz[1] || yz[1];
    const bool clear = xc[1] || (xb[1] && yc[1]);
    od = zero ? 0.0f : (clear ? 1.0f : ds[k]);
    oc = od < 1.0f ? make_float4(0.0f, 0.0f, 0.0f, -1.0f) : fc_clear_col();
  } else {
    depth_av /= (float)num;
    float tr = 0.0f, tg = 0.0f, tb = 0.0f, td = 0.0f, tw = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const bool sel = live[k] && cs[k].x >= 0.0f && ds[k] >= depth_av;
      tr += sel ? cs[k].x * 1.0f : 0.0f;
      tg += sel ? cs[k].y * 1.0f : 0.0f;
      tb += sel ? cs[k].z * 1.0f : 0.0f;
      td += sel ? ds[k] * 1.0f : 0.0f;
      tw += sel ? 1.0f : 0.0f;
    }
    oc = make_float4(tr / tw, tg / tw, tb / tw, 1.0f);
    od = td / tw;
  }
}

// LOD i of the band from the state before it, 32 x 8 texels per block.  TABLES: the tap positions come from the tables
// (LOD 1, where the arithmetic of 230 000 texels counts); otherwise they are computed (the LODs in between: a few
// thousand texels whose launch lasts as long as one texel's chain of dependent loads and instructions).  With
// `clear_below` the launch also covers the band's rows under LOD i's viewport and sets them to the clear colour (i = 1:
// the rows every later LOD lives in); clear_below == 2 does only that.
template <bool TABLES>
__global__ void __launch_bounds__(256) k_fc_inpaint(FillLayout L, FillTabs T, int i, FillSrc S, float4* __restrict__ acol,
                                                    float* __restrict__ adep, int clear_below)
{
  const int AW = L.FW - L.W;
  const int fx = blockIdx.x * 32 + (threadIdx.x & 31);
  int row = blockIdx.y * 8 + (threadIdx.x >> 5);
  if (clear_below) {  // rows [0, off.y) of the band first, then the viewport
    if (row < L.off[i][1]) {
      if (fx < AW) {
        acol[(size_t)row * AW + fx] = fc_clear_col();
        adep[(size_t)row * AW + fx] = 1.0f;
      }
      return;
    }
    if (clear_below == 2) return;
    row -= L.off[i][1];
  }
  const int fy = row;
  if (fx >= L.res[i][0] || fy >= L.res[i][1]) return;
  int xe[4], ye[4];
  if (TABLES) {
    const int4 xq = T.xt[T.xbase[i] + fx], yq = T.yt[T.ybase[i] + fy];
    xe[0] = xq.x, xe[1] = xq.y, xe[2] = xq.z, xe[3] = xq.w;
    ye[0] = yq.x, ye[1] = yq.y, ye[2] = yq.z, ye[3] = yq.w;
  } else {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      xe[t] = fc_tap_column(L, i, fx, t);
      ye[t] = fc_tap_row(L, i, fy, t);
    }
  }
  float4 oc;
  float od;
  fc_inpaint_texel<false>(L, S, xe, ye, FillLds{nullptr, nullptr, 0, 0, 0, 0}, oc, od);
  const size_t o = (size_t)(L.off[i][1] + fy) * AW + (L.off[i][0] - L.W + fx);
  acol[o] = oc;
  adep[o] = od;
}

// The tail of the pyramid, LODs [first, num_lods), in ONE workgroup: what its passes read of N is copied to LDS once
// (FillLds), every LOD it computes goes there for the LODs after it (and to the band for colorfill), its tap tables too;
// the passes are separated by __syncthreads() only and touch global memory for their stores alone.
__global__ void __launch_bounds__(1024) k_fc_inpaint_tail(FillLayout L, FillTabs T, int first, FillSrc S, float4* __restrict__ acol,
                                                          float* __restrict__ adep, int cap)
{
  extern __shared__ int4 fc_lds[];
  const int ntx = T.nx - T.xbase[first], nty = T.ny - T.ybase[first];
  int4* lxt = fc_lds;
  int4* lyt = lxt + ntx;
  float4* lcol = (float4*)(lyt + nty);
  float* ldep = (float*)(lcol + cap);
  const int y1 = L.off[first][1] + L.res[first][1];  // first row above the tail's LODs
  const FillLds M{lcol, ldep, L.W - 3, L.off[L.num_lods - 1][1], L.res[first][0] + 6, y1 + 3 - L.off[L.num_lods - 1][1]};
  const int AW = L.FW - L.W;
  for (int t = threadIdx.x; t < ntx; t += blockDim.x) lxt[t] = T.xt[T.xbase[first] + t];
  for (int t = threadIdx.x; t < nty; t += blockDim.x) lyt[t] = T.yt[T.ybase[first] + t];
  for (int t = threadIdx.x; t < cap; t += blockDim.x) {
    const int y = M.y0 + t / M.w, x = M.x0 + t % M.w;
    float4 c = fc_clear_col();  // the band under the LODs computed before this launch
    float d = 1.0f;
    if (x >= 0 && x < L.FW && y < L.H && (x < L.W || y >= y1)) fc_fetch(L, S, x, y, c, d);
    lcol[t] = c;
    ldep[t] = d;
  }
  __syncthreads();
  for (int i = first; i < L.num_lods; ++i) {
    const int rx = L.res[i][0], n = rx * L.res[i][1];
    const int xb0 = T.xbase[i] - T.xbase[first], yb0 = T.ybase[i] - T.ybase[first];
    for (int t = threadIdx.x; t < n; t += blockDim.x) {
      const int fy = t / rx, fx = t - fy * rx;
      const int4 xq = lxt[xb0 + fx], yq = lyt[yb0 + fy];
      const int xe[4] = {xq.x, xq.y, xq.z, xq.w}, ye[4] = {yq.x, yq.y, yq.z, yq.w};
      float4 oc;
      float od;
      fc_inpaint_texel<true>(L, S, xe, ye, M, oc, od);
      const int lt = (L.off[i][1] + fy - M.y0) * M.w + (L.off[i][0] + fx - M.x0);
      lcol[lt] = oc;
      ldep[lt] = od;
      const size_t o = (size_t)(L.off[i][1] + fy) * AW + (L.off[i][0] - L.W + fx);
      acol[o] = oc;
      adep[o] = od;
    }
    __syncthreads();
  }
}

// MIRRORED_REPEAT: mirror((i mod 2n), n); the coordinates colorfill builds lie within a texel of the atlas, where the
// reflection needs no division
__device__ __forceinline__ int fc_mirror(int i, int n)
{
  if (i >= -n && i < 2 * n) return i < 0 ? -1 - i : (i >= n ? 2 * n - 1 - i : i);
  const int period = 2 * n;
  int k = i % period;
  if (k < 0) k += period;
  return k < n ? k : period - 1 - k;
}

__device__ __forceinline__ float4 fc_texel(const FillLayout& L, const FillSrc& S, int x, int y)
{
  const bool band = x >= L.W;
  const unsigned at = band ? (unsigned)(y * (L.FW - L.W) + (x - L.W)) : (unsigned)(y * L.W + x);
  return (band ? S.acol : S.fcol)[at];
}

// texture(texture_color, p): LINEAR + MIRRORED_REPEAT (view_lod.cpp:52-53)
__device__ __forceinline__ float4 fc_texture(const FillLayout& L, const FillSrc& S, float u, float v)
{
  const int FW = L.FW, H = L.H;
  const float tx = u * (float)FW - 0.5f, ty = v * (float)H - 0.5f;
  const float fx = floorf(tx), fy = floorf(ty);
  const float ax = tx - fx, ay = ty - fy;
  const int jx = idx_from_floor(fx, FW * 4), jy = idx_from_floor(fy, H * 4);
  const int x0 = fc_mirror(jx, FW), x1 = fc_mirror(jx + 1, FW), y0 = fc_mirror(jy, H), y1 = fc_mirror(jy + 1, H);
  const float4 t00 = fc_texel(L, S, x0, y0), t10 = fc_texel(L, S, x1, y0);
  const float4 t01 = fc_texel(L, S, x0, y1), t11 = fc_texel(L, S, x1, y1);
  return make_float4(lerpf(lerpf(t00.x, t10.x, ax), lerpf(t01.x, t11.x, ax), ay), lerpf(lerpf(t00.y, t10.y, ax), lerpf(t01.y, t11.y, ax), ay),
                     lerpf(lerpf(t00.z, t10.z, ax), lerpf(t01.z, t11.z, ax), ay), lerpf(lerpf(t00.w, t10.w, ax), lerpf(t01.w, t11.w, ax), ay));
}

__device__ __forceinline__ float fc_clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

// tsdf_colorfill.fs into the W x H output.  A pixel the ray-march hit (alpha > 0 in LOD 0) is a copy.  For a hole only the
// ALPHA of the LODs decides the level (the colour found there is replaced by the blend of the two LODs after it), so the
// level loop is one round of 4-byte loads instead of a chain of dependent texel fetches; the offsets / resolutions of the
// two LODs a lane blends are read from a table in LDS (a per-lane index into the kernel arguments is a select chain).
__global__ void __launch_bounds__(256) k_fc_colorfill(FillLayout L, FillSrc S, float rix, float riy, float4* __restrict__ out_col,
                                                      float* __restrict__ out_dep)
{
  __shared__ float4 lodf[24];  // (off.x, off.y, res.x, res.y) of LOD l; 0 from num_lods on (the shader's uniform arrays)
  if (threadIdx.x < 24) {
    const int l = threadIdx.x;
    lodf[l] = l < L.num_lods ? make_float4((float)L.off[l][0], (float)L.off[l][1], (float)L.res[l][0], (float)L.res[l][1])
                             : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  }
  __syncthreads();
  const int px = blockIdx.x * blockDim.x + threadIdx.x, py = blockIdx.y;
  if (px >= L.W) return;
  const float tcx = (float)px / (float)L.res[0][0], tcy = (float)py / (float)L.res[0][1];
  float4 c;
  float d0;
  fc_fetch(L, S, (int)((float)L.off[0][0] + (float)L.res[0][0] * tcx), (int)((float)L.off[0][1] + (float)L.res[0][1] * tcy), c, d0);
  if (!(c.w > 0.0f)) {
    const int AW = L.FW - L.W;
    int level = L.num_lods;
    for (int l = L.num_lods - 1; l >= 1; --l) {  // LODs >= 1 lie in the band (x offset W); the loads do not depend on each other
      const int cx = (int)((float)L.off[l][0] + (float)L.res[l][0] * tcx);
      const int cy = (int)((float)L.off[l][1] + (float)L.res[l][1] * tcy);
      // (at widths where W + res * tcx rounds up to the atlas' edge the fetch leaves the texture: 0)
      const bool inside = cx < L.FW && cy < L.H;
      const float alpha = S.acol[inside ? (unsigned)(cy * AW + (cx - L.W)) : 0u].w;
      if (inside && alpha > 0.0f) level = l;
    }
    const float ptx = ((float)px + 0.5f) / (float)L.W, pty = ((float)py + 0.5f) / (float)L.H;
    const float4 la = lodf[level + 1], lb = lodf[level + 2];
    const float p0x = fc_clampf(la.x + la.z * ptx, la.x + 0.5f, (la.x + la.z) - 0.5f) * rix;
    const float p0y = fc_clampf(la.y + la.w * pty, la.y + 0.5f, (la.y + la.w) - 0.5f) * riy;
    const float p1x = fc_clampf(lb.x + lb.z * ptx, lb.x + 0.5f, (lb.x + lb.z) - 0.5f) * rix;
    const float p1y = fc_clampf(lb.y + lb.w * pty, lb.y + 0.5f, (lb.y + lb.w) - 0.5f) * riy;
    const float4 c1 = fc_texture(L, S, p0x, p0y);
    const float4 c2 = fc_texture(L, S, p1x, p1y);
    const float w1 = sqrtf(ptx * ptx + pty * pty);
    const float w2 = 1.0f - w1;
    c = make_float4((c1.x * w1 + c2.x * w2) / (w1 + w2), (c1.y * w1 + c2.y * w2) / (w1 + w2), (c1.z * w1 + c2.z * w2) / (w1 + w2),
                    (c1.w * w1 + c2.w * w2) / (w1 + w2));
  }
  out_col[(size_t)py * L.W + px] = c;
  out_dep[(size_t)py * L.W + px] = d0;
}

// texels of the first LOD the tail workgroup takes, and of the copy of N it may hold in LDS (20 B each, under 64 KiB with
// the tables)
static constexpr int FC_TAIL_TEXELS = 1024, FC_TAIL_LDS_TEXELS = 3000;

size_t fill_band_texels(const FillLayout& L) { return (size_t)(L.FW - L.W) * L.H; }

static int fc_tail_cap(const FillLayout& L, int first)
{
  return (L.res[first][0] + 6) * (L.off[first][1] + L.res[first][1] + 3 - L.off[L.num_lods - 1][1]);
}

void launch_fill_colors(const FillLayout& L, const FillTabs& T, const float4* frame_col, const float* frame_dep, float4* acol,
                        float* adep, float4* out_col, float* out_dep, hipStream_t s)
{
  const FillSrc S{frame_col, frame_dep, acol, adep};
  int first = L.num_lods;  // first LOD of the tail
  for (int i = 1; i < L.num_lods; ++i)
    if (L.res[i][0] * (long long)L.res[i][1] <= FC_TAIL_TEXELS && fc_tail_cap(L, i) <= FC_TAIL_LDS_TEXELS) {
      first = i;
      break;
    }
  for (int i = 1; i < first; ++i) {
    const int rows = L.res[i][1] + (i == 1 ? L.off[1][1] : 0);
    const dim3 grid((L.res[i][0] + 31) / 32, (rows + 7) / 8);
    if (i == 1)
      hipLaunchKernelGGL(k_fc_inpaint<true>, grid, dim3(256), 0, s, L, T, i, S, acol, adep, 1);
    else
      hipLaunchKernelGGL(k_fc_inpaint<false>, grid, dim3(256), 0, s, L, T, i, S, acol, adep, 0);
  }
  if (first < L.num_lods) {
    if (first == 1)  // a frame so small that no launch above cleared the band
      hipLaunchKernelGGL(k_fc_inpaint<false>, dim3((L.FW - L.W + 31) / 32, (L.off[1][1] + 7) / 8), dim3(256), 0, s, L, T, 1, S, acol, adep, 2);
    const int cap = fc_tail_cap(L, first);
    const size_t lds = (size_t)cap * 20 + (size_t)(T.nx - T.xbase[first] + T.ny - T.ybase[first]) * 16;
    hipLaunchKernelGGL(k_fc_inpaint_tail, dim3(1), dim3(1024), lds, s, L, T, first, S, acol, adep, cap);
  }
  hipLaunchKernelGGL(k_fc_colorfill, dim3((L.W + 255) / 256, L.H), dim3(256), 0, s, L, S, 1.0f / (float)L.FW, 1.0f / (float)L.H, out_col,
                     out_dep);
}

}  // namespace rgbdr
