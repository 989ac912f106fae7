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
// A texel is computed by a QUAD of lanes (fc_inpaint_quad): a pass of a few hundred texels lasts as long as one lane's
// instruction stream, and a dependent launch costs 1.7 us (queue filled ahead) to 3.3 us (host-bound) before it does anything
// (profiles/probes_src/launch_floor_probe.hip).
// Launches per frame: LOD 1 (+ clear of the band below it), one per LOD with more texels than the tail's first, ONE
// workgroup for the tail (its LODs live in LDS between __syncthreads(), no global-memory phase), colorfill.
// 1280 x 720: 21 -> 7 launches, 0.122 -> 0.062 ms.
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

// ---- one texel on a quad of lanes ---------------------------------------------------------------------------------
// A wavefront with nothing beside it on its SIMD issues one instruction per ~7 cycles (profiles/probes_src/
// launch_floor_probe.hip), so a pass of a few hundred texels lasts as long as ONE texel's instruction stream: 1200
// instructions per texel were 4-6 us per LOD whatever its size.  Four lanes share a texel -- lane q owns tap row q (its four
// taps, loads and selects) -- and the shader's two order-dependent sums run as chains over the quad (DPP quad_perm, no LDS):
//   depth_av, order x outer / y inner: step (x, y) happens in lane y: acc = acc of lane y - 1 + s[x];
//   the colour / depth totals, order k = x + 4 y: round y happens in lane y: four local adds onto lane y - 1's totals.
// Every lane executes every step; only the lane whose turn it is continues the chain, the others' results are never read.
template <int CTRL>
__device__ __forceinline__ float fc_dpp(float v)
{
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ int fc_dpp(int v)
{
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}
constexpr int fc_quad(int a, int b, int c, int d) { return a | (b << 2) | (c << 4) | (d << 6); }
constexpr int FC_FROM_PREV = fc_quad(3, 0, 1, 2);  // lane q reads lane q - 1 (lane 0: lane 3)
constexpr int FC_XOR1 = fc_quad(1, 0, 3, 2), FC_XOR2 = fc_quad(2, 3, 0, 1);
template <class T>
__device__ __forceinline__ T fc_quad_bcast(T v, int from)  // `from` is a compile-time constant at every call
{
  switch (from) {
    case 0: return fc_dpp<fc_quad(0, 0, 0, 0)>(v);
    case 1: return fc_dpp<fc_quad(1, 1, 1, 1)>(v);
    case 2: return fc_dpp<fc_quad(2, 2, 2, 2)>(v);
    default: return fc_dpp<fc_quad(3, 3, 3, 3)>(v);
  }
}
__device__ __forceinline__ int fc_quad_sum(int v)
{
  v += fc_dpp<FC_XOR1>(v);
  return v + fc_dpp<FC_XOR2>(v);
}

// tsdf_inpaint.fs for one texel of LOD i on the four lanes of a quad (q = lane & 3): xq = N column of tap column q, yq = N
// row of tap row q (fill_taps.cuh), the native atlas read in its state before this pass.  Lane q returns component q of
// out_FragColor in `comp`; lane 3 also gl_FragDepth in `od`.  A tap outside the atlas or on the clear colour has alpha 0:
// it takes part in nothing but the depth of the num_samples == 0 branch (centre tap).  LDS: taps inside M come from there,
// and the global loads are issued only if some lane of the wavefront has a tap elsewhere.
template <bool LDS>
__device__ __forceinline__ void fc_inpaint_quad(const FillLayout& L, const FillSrc& S, int q, int xq, int yq, const FillLds& M,
                                                float& comp, float& od)
{
  const int AW = L.FW - L.W;
  int xe[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) xe[t] = fc_quad_bcast(xq, t);
  const bool yz = yq < 0;
  const bool yc = !yz && (yq & FC_ROW_CLEAR) != 0;
  const int ny = yz ? 0 : (yq & ~FC_ROW_CLEAR);
  const bool yl = LDS && !yz && ny >= M.y0 && ny < M.y0 + M.h;
  bool live[4], inl[4], band[4], special1[4];
  unsigned at[4];
  int lt[4];
  bool need_global = !LDS;
#pragma unroll
  for (int x = 0; x < 4; ++x) {
    const bool xz = xe[x] == FC_TAP_OUTSIDE, xc = xe[x] == FC_TAP_CLEAR;
    band[x] = xe[x] >= L.W;
    const int colv = xe[x] < 0 ? 0 : (band[x] ? xe[x] - L.W : xe[x]);
    const bool zero = xz || yz;
    const bool clear = xc || (band[x] && yc);
    special1[x] = !zero && clear;  // reads as the clear colour (depth 1); zero && ... reads as 0
    live[x] = !zero && !clear;
    at[x] = (unsigned)(ny * (band[x] ? AW : L.W) + colv);  // a texel of the frame / band whatever the kind
    inl[x] = LDS && live[x] && yl && xe[x] >= M.x0 && xe[x] < M.x0 + M.w;
    lt[x] = inl[x] ? (ny - M.y0) * M.w + (xe[x] - M.x0) : 0;
    if (LDS) need_global = need_global || (live[x] && !inl[x]);
  }
  float4 cs[4];
  float ds[4];
  if (!LDS || __builtin_amdgcn_ballot_w64(need_global) != 0) {
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      cs[x] = (band[x] ? S.acol : S.fcol)[at[x]];
      ds[x] = (band[x] ? S.adep : S.fdep)[at[x]];
    }
  } else {
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      cs[x] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      ds[x] = 0.0f;
    }
  }
  if (LDS) {
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      const float4 lc = M.col[lt[x]];
      const float ld = M.dep[lt[x]];
      if (inl[x]) {
        cs[x] = lc;
        ds[x] = ld;
      }
    }
  }
  // a sum that starts at +0 is never -0, so adding +0 for a tap that does not count leaves the shader's sum bit for bit
  float s[4];
  int cnt = 0;
#pragma unroll
  for (int x = 0; x < 4; ++x) {
    live[x] = live[x] && !(cs[x].w <= 0.0f);
    s[x] = live[x] ? ds[x] : 0.0f;
    cnt += live[x] ? 1 : 0;
  }
  float acc = 0.0f;
#pragma unroll
  for (int x = 0; x < 4; ++x)  // the shader's order of accumulation: x outer, y inner -- step (x, y) in lane y
#pragma unroll
    for (int y = 0; y < 4; ++y) acc = fc_dpp<FC_FROM_PREV>(acc) + s[x];
  const int num = fc_quad_sum(cnt);
  if (num == 0) {  // (the same in the four lanes)
    // texelFetch(texture_depth, pos_int): tap (1, 1), lane 1
    const float dm = fc_quad_bcast((xe[1] == FC_TAP_OUTSIDE || yz) ? 0.0f : (special1[1] ? 1.0f : ds[1]), 1);
    od = dm;
    comp = dm < 1.0f ? (q == 3 ? -1.0f : 0.0f) : (q == 1 ? 1.0f : 0.0f);
    return;
  }
  const float depth_av = fc_quad_bcast(acc, 3) / (float)num;
  float tr = 0.0f, tg = 0.0f, tb = 0.0f, td = 0.0f;
  float v[4][4];
  int nsel = 0;
#pragma unroll
  for (int x = 0; x < 4; ++x) {
    const bool sel = live[x] && cs[x].x >= 0.0f && ds[x] >= depth_av;
    v[x][0] = sel ? cs[x].x * 1.0f : 0.0f;
    v[x][1] = sel ? cs[x].y * 1.0f : 0.0f;
    v[x][2] = sel ? cs[x].z * 1.0f : 0.0f;
    v[x][3] = sel ? ds[x] * 1.0f : 0.0f;
    nsel += sel ? 1 : 0;
  }
#pragma unroll
  for (int y = 0; y < 4; ++y) {  // order k = x + 4 y: round y in lane y
    tr = fc_dpp<FC_FROM_PREV>(tr) + v[0][0];
    tg = fc_dpp<FC_FROM_PREV>(tg) + v[0][1];
    tb = fc_dpp<FC_FROM_PREV>(tb) + v[0][2];
    td = fc_dpp<FC_FROM_PREV>(td) + v[0][3];
#pragma unroll
    for (int x = 1; x < 4; ++x) {
      tr += v[x][0];
      tg += v[x][1];
      tb += v[x][2];
      td += v[x][3];
    }
  }
  const float tw = (float)fc_quad_sum(nsel);  // a sum of ones: exact
  // lane 3 holds the totals; lane q divides the q-th
  const float r3 = fc_quad_bcast(tr, 3), g3 = fc_quad_bcast(tg, 3), b3 = fc_quad_bcast(tb, 3), d3 = fc_quad_bcast(td, 3);
  const float mine = (q == 0 ? r3 : (q == 1 ? g3 : (q == 2 ? b3 : d3))) / tw;
  od = mine;                      // lane 3: total_depth / total_weight
  comp = q == 3 ? 1.0f : mine;    // vec4(total_color / total_weight, 1)
}

// one texel's result from its quad into colour / depth arrays at texel `o`
__device__ __forceinline__ void fc_store_quad(float4* col, float* dep, size_t o, int q, float comp, float od)
{
  ((float*)(col + o))[q] = comp;
  if (q == 3) dep[o] = od;
}

// LOD i of the band from the state before it: blocks of 16 x 4 texels x 4 lanes.  TABLES: the tap positions come from the
// tables; otherwise each lane computes its column and its row (a cold table load costs a small launch more than the
// arithmetic).  With `clear_rows` the launch also sets the band's rows under LOD i's viewport -- one contiguous range -- to
// the clear colour in the blocks beyond `tiles_y` (i = 1: the rows every later LOD lives in).
template <bool TABLES>
__global__ void __launch_bounds__(256) k_fc_inpaint(FillLayout L, FillTabs T, int i, FillSrc S, float4* __restrict__ acol,
                                                    float* __restrict__ adep, int tiles_y)
{
  const int AW = L.FW - L.W;
  if ((int)blockIdx.y >= tiles_y) {
    const size_t n = (size_t)L.off[i][1] * AW, stride = (size_t)gridDim.x * (gridDim.y - tiles_y) * 256;
    for (size_t t = ((size_t)(blockIdx.y - tiles_y) * gridDim.x + blockIdx.x) * 256 + threadIdx.x; t < n; t += stride) {
      acol[t] = fc_clear_col();
      adep[t] = 1.0f;
    }
    return;
  }
  const int q = threadIdx.x & 3, texel = threadIdx.x >> 2;
  const int fx = blockIdx.x * 16 + (texel & 15), fy = blockIdx.y * 4 + (texel >> 4);
  if (fx >= L.res[i][0] || fy >= L.res[i][1]) return;  // (a whole quad)
  int xq, yq;
  if (TABLES) {
    xq = ((const int*)T.xt)[(T.xbase[i] + fx) * 4 + q];
    yq = ((const int*)T.yt)[(T.ybase[i] + fy) * 4 + q];
  } else {
    xq = fc_tap_column(L, i, fx, q);
    yq = fc_tap_row(L, i, fy, q);
  }
  float comp, od;
  fc_inpaint_quad<false>(L, S, q, xq, yq, FillLds{nullptr, nullptr, 0, 0, 0, 0}, comp, od);
  fc_store_quad(acol, adep, (size_t)(L.off[i][1] + fy) * AW + (L.off[i][0] - L.W + fx), q, comp, od);
}

// The tail of the pyramid, LODs [first, num_lods), in ONE workgroup: what its passes read of N is copied to LDS once
// (FillLds), every LOD it computes goes there for the LODs after it (and to the band for colorfill), its tap tables too;
// the passes are separated by __syncthreads() only and touch global memory for their stores alone.
__global__ void __launch_bounds__(1024) k_fc_inpaint_tail(FillLayout L, FillTabs T, int first, FillSrc S, float4* __restrict__ acol,
                                                          float* __restrict__ adep, int cap)
{
  extern __shared__ int4 fc_lds[];
  const int ntx = T.nx - T.xbase[first], nty = T.ny - T.ybase[first];
  int* lxt = (int*)fc_lds;
  int* lyt = lxt + 4 * ntx;
  float4* lcol = (float4*)(lyt + 4 * nty);
  float* ldep = (float*)(lcol + cap);
  const int y1 = L.off[first][1] + L.res[first][1];  // first row above the tail's LODs
  const FillLds M{lcol, ldep, L.W - 3, L.off[L.num_lods - 1][1], L.res[first][0] + 6, y1 + 3 - L.off[L.num_lods - 1][1]};
  const int AW = L.FW - L.W;
  for (int t = threadIdx.x; t < ntx; t += blockDim.x) ((int4*)lxt)[t] = T.xt[T.xbase[first] + t];
  for (int t = threadIdx.x; t < nty; t += blockDim.x) ((int4*)lyt)[t] = T.yt[T.ybase[first] + t];
  for (int t = threadIdx.x; t < cap; t += blockDim.x) {
    const int y = M.y0 + t / M.w, x = M.x0 + t % M.w;
    float4 c = fc_clear_col();  // the band under the LODs computed before this launch
    float d = 1.0f;
    if (x >= 0 && x < L.FW && y < L.H && (x < L.W || y >= y1)) fc_fetch(L, S, x, y, c, d);
    lcol[t] = c;
    ldep[t] = d;
  }
  __syncthreads();
  const int q = threadIdx.x & 3;
  for (int i = first; i < L.num_lods; ++i) {
    const int rx = L.res[i][0], n = rx * L.res[i][1];
    const int xb0 = T.xbase[i] - T.xbase[first], yb0 = T.ybase[i] - T.ybase[first];
    for (int t = threadIdx.x >> 2; t < n; t += blockDim.x >> 2) {  // (a quad per texel)
      const int fy = t / rx, fx = t - fy * rx;
      float comp, od;
      fc_inpaint_quad<true>(L, S, q, lxt[(xb0 + fx) * 4 + q], lyt[(yb0 + fy) * 4 + q], M, comp, od);
      fc_store_quad(lcol, ldep, (size_t)((L.off[i][1] + fy - M.y0) * M.w + (L.off[i][0] + fx - M.x0)), q, comp, od);
      fc_store_quad(acol, adep, (size_t)(L.off[i][1] + fy) * AW + (L.off[i][0] - L.W + fx), q, comp, od);
    }
    __syncthreads();
  }
}

// MIRRORED_REPEAT: mirror(i mod 2n, n).  The coordinates colorfill builds lie within two texels of the atlas, where the
// reflection is ~i for i < 0 and min(i, 2n - 1 - i) above; anything else takes the division.
__device__ __forceinline__ int fc_mirror(int i, int n)
{
  if (i < -n || i >= 2 * n) {
    const int period = 2 * n;
    int k = i % period;
    if (k < 0) k += period;
    return k < n ? k : period - 1 - k;
  }
  const int m = i ^ (i >> 31);  // -1 - i for i < 0
  return min(m, 2 * n - 1 - m);
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

// texels of the first LOD the tail workgroup takes (a quad of lanes each: one round of its 1024 threads), and of the copy
// of N it may hold in LDS (20 B each, under 64 KiB with the tables)
static constexpr int FC_TAIL_TEXELS = 256, FC_TAIL_LDS_TEXELS = 3000;

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
    const int tiles_x = (L.res[i][0] + 15) / 16, tiles_y = (L.res[i][1] + 3) / 4;
    if (i == 1) {  // + blocks that clear the rows under the viewport
      const long long clear_blocks = ((long long)L.off[1][1] * (L.FW - L.W) + 1023) / 1024;
      const int extra_y = (int)((clear_blocks + tiles_x - 1) / tiles_x);
      hipLaunchKernelGGL(k_fc_inpaint<true>, dim3(tiles_x, tiles_y + extra_y), dim3(256), 0, s, L, T, i, S, acol, adep, tiles_y);
    } else {
      hipLaunchKernelGGL(k_fc_inpaint<false>, dim3(tiles_x, tiles_y), dim3(256), 0, s, L, T, i, S, acol, adep, tiles_y);
    }
  }
  if (first < L.num_lods) {
    if (first == 1) {  // a frame so small that no launch above cleared the band
      const int blocks = (int)(((long long)L.off[1][1] * (L.FW - L.W) + 1023) / 1024);
      if (blocks > 0) hipLaunchKernelGGL(k_fc_inpaint<false>, dim3(1, blocks), dim3(256), 0, s, L, T, 1, S, acol, adep, 0);
    }
    const int cap = fc_tail_cap(L, first);
    const size_t lds = (size_t)cap * 20 + (size_t)(T.nx - T.xbase[first] + T.ny - T.ybase[first]) * 16;
    hipLaunchKernelGGL(k_fc_inpaint_tail, dim3(1), dim3(1024), lds, s, L, T, first, S, acol, adep, cap);
  }
  hipLaunchKernelGGL(k_fc_colorfill, dim3((L.W + 255) / 256, L.H), dim3(256), 0, s, L, S, 1.0f / (float)L.FW, 1.0f / (float)L.H, out_col,
                     out_dep);
}

}  // namespace rgbdr
