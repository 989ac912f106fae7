// integrate_fold.cuh -- what the sweeps of the TSDF share (kernels_integrate.hip: full sweep; kernels_bricks.hip:
// brick-skipping sweep; kernels_skip.hip: background-skip sweep): the packed frame texel, one sensor's fold of one
// voxel (tsdf_integration.vs:31-54) from an LDS window, the reference's brick -> voxel membership, and the body that
// loads and folds one sensor group of one tile.
#pragma once
#include <hip/hip_runtime.h>

#include "rgbdr_internal.hpp"
#include "sampling.cuh"

namespace rgbdr {

// Block-uniform words that no kernel writes while the sweep runs (tables, window origins, the list) are read
// through the constant address space: scalar loads.  As plain global loads they become vector loads that are
// waited for with vmcnt(0), which would drain the prefetched stage every time.
template <class T>
__device__ __forceinline__ __attribute__((address_space(4))) const T* ro(const T* p)
{
  return (__attribute__((address_space(4))) const T*)p;
}

constexpr int kWin = 16;      // frame window edge staged in LDS per (tile, sensor)
// LDS row pitch of a window in texels.  16 texels would be 128 B = all 32 banks, so
// footprints in the same column of different rows would collide (measured: 79 % of
// the LDS cycles were bank-conflict cycles); 17 rotates each row by two banks.
constexpr int kWinPitch = 17;

// packed frame texel (kernels_pre.hip k_quality): x = depth_b.r, y = quality with
// "silhouette == 0" in the sign bit
__device__ __forceinline__ float texel_sil(uint2 t) { return (t.y >> 31) ? 0.0f : 1.0f; }
__device__ __forceinline__ float texel_quality(uint2 t) { return __uint_as_float(t.y & 0x7fffffffu); }
__device__ __forceinline__ float texel_depth(uint2 t) { return __uint_as_float(t.x); }

// One sensor's contribution to one voxel (tsdf_integration.vs:31-54) from the
// 2x2 LINEAR footprint of (pcx, pcy) in that sensor's frame.
__device__ __forceinline__ void fold_taps(uint2 p00, uint2 p10, uint2 p01, uint2 p11, float ax, float ay, float pcz,
                                          float limit, float& tsd, float& wsum)
{
  const float sil = lerpf(lerpf(texel_sil(p00), texel_sil(p10), ax), lerpf(texel_sil(p01), texel_sil(p11), ax), ay);
  if (sil < 1.0f && tsd >= limit) {
    tsd = -limit;
    return;
  }
  // NEAREST depth texel = floor(s*n) = j + (a >= 0.5) on each axis (a = frac(s*n - 0.5)):
  // always one of the four texels of the LINEAR footprint, index clamping included
  const uint2 n0 = (ax >= 0.5f) ? p10 : p00;
  const uint2 n1 = (ax >= 0.5f) ? p11 : p01;
  const float depth = texel_depth((ay >= 0.5f) ? n1 : n0);
  const float sdist = pcz - depth;
  if (sdist <= -limit) {
    tsd = -limit;
  } else if (sdist >= limit) {
  } else {
    const float weight =
        lerpf(lerpf(texel_quality(p00), texel_quality(p10), ax), lerpf(texel_quality(p01), texel_quality(p11), ax), ay);
    tsd = (tsd * wsum + weight * sdist) / (wsum + weight);
    wsum += weight;
  }
}

// footprint position of a normalised coordinate: j = floor(s*n - 0.5) saturated to
// [-1, n], a = fraction (same arithmetic as axis_linear)
__device__ __forceinline__ int footprint(float s, int n, float& a)
{
  const float t = s * (float)n - 0.5f;
  const float f = floorf(t);
  a = t - f;
  return idx_from_floor(f, n);
}

// global-memory footprint fetch with CLAMP_TO_EDGE
__device__ __forceinline__ void fetch_global(const uint2* __restrict__ frame, int W, int H, int jx, int jy, uint2& p00,
                                             uint2& p10, uint2& p01, uint2& p11)
{
  const int x0 = clampi(jx, 0, W - 1), x1 = clampi(jx + 1, 0, W - 1);
  const int r0 = clampi(jy, 0, H - 1) * W, r1 = clampi(jy + 1, 0, H - 1) * W;
  p00 = frame[r0 + x0];
  p10 = frame[r0 + x1];
  p01 = frame[r1 + x0];
  p11 = frame[r1 + x1];
}

// `win` must point into LDS.  The LDS reads are unconditional (safe cell 0 when the
// footprint is outside the window) so they stay ds_read instructions; the global
// fetch is a rare, separate branch.
__device__ __forceinline__ void fold_voxel_window(const uint2* win, int wx0, int wy0,
                                                  const uint2* __restrict__ frame, int W, int H, float pcx, float pcy,
                                                  float pcz, float limit, float& tsd, float& wsum)
{
  float ax, ay;
  const int jx = footprint(pcx, W, ax), jy = footprint(pcy, H, ay);
  const int rx = jx - wx0, ry = jy - wy0;
  const bool inside = (unsigned)rx < (unsigned)(kWin - 1) && (unsigned)ry < (unsigned)(kWin - 1);
  const int cell = inside ? ry * kWinPitch + rx : 0;
  // explicit LDS address space: keeps these ds_read2_b64 (a generic pointer merged
  // with the global fallback would turn all eight loads into flat_load)
  typedef __attribute__((address_space(3))) const unsigned long long lds_texel;
  lds_texel* w = (lds_texel*)win + cell;
  const unsigned long long t00 = w[0], t10 = w[1], t01 = w[kWinPitch], t11 = w[kWinPitch + 1];
  uint2 p00 = make_uint2((unsigned)t00, (unsigned)(t00 >> 32)), p10 = make_uint2((unsigned)t10, (unsigned)(t10 >> 32));
  uint2 p01 = make_uint2((unsigned)t01, (unsigned)(t01 >> 32)), p11 = make_uint2((unsigned)t11, (unsigned)(t11 >> 32));
  if (__builtin_expect(!inside, 0)) {  // invalid LUT entry, tile close to the sensor ...
    fetch_global(frame, W, H, jx, jy, p00, p10, p01, p11);
  }
  fold_taps(p00, p10, p01, p11, ax, ay, pcz, limit, tsd, wsum);
}

// Any occupied brick among those that hold the index triple (xs, ys, zs) of
// VolumeSampler::containedVoxels (volume_sampler.cpp:53-55).  Membership is separable per
// axis (BrickTables, geometry.cpp); an index past the x / y end lies in the last brick there.
__device__ __forceinline__ bool bricks_any(const IntegrateParams& p, int xs, int ys, int zs)
{
  const uint32_t ex = xs < p.X ? p.vbx[xs] : (uint32_t)(p.bx - 1) * 0x10001u;
  const uint32_t ey = ys < p.Y ? p.vby[ys] : (uint32_t)(p.by - 1) * 0x10001u;
  const uint32_t ez = p.vbz[zs];
  bool any = false;
  for (uint32_t bz = ez & 0xffffu; bz <= (ez >> 16); ++bz)
    for (uint32_t by = ey & 0xffffu; by <= (ey >> 16); ++by)
      for (uint32_t bx = ex & 0xffffu; bx <= (ex >> 16); ++bx) any |= p.brick_mask[((size_t)bz * p.by + by) * p.bx + bx] != 0;
  return any;
}

// Is voxel (vx, vy, vz) in the index list of an occupied brick (recon_integration.cpp:255-259)?
// The lists hold linear indices z*X*Y + y*X + x (volume_sampler.cpp:57); where the last brick of
// the x or y axis reaches `ovx` / `ovy` indices past the axis end, those indices alias voxels of
// the next row / slice, so up to four index triples produce this voxel's linear index.  Indices
// past the z end leave the vertex buffer and are dropped.
__device__ __forceinline__ bool voxel_occupied(const IntegrateParams& p, int vx, int vy, int vz)
{
  if (vx >= p.X || vy >= p.Y || vz >= p.Z) return false;  // padding voxel of a partial tile
  bool any = bricks_any(p, vx, vy, vz);
  if (__builtin_expect((p.ovx | p.ovy) != 0, 0)) {
    for (int kx = 0; kx < 2; ++kx) {
      if (kx == 1 && vx >= p.ovx) break;
      const int m = vz * p.Y + vy - kx;  // ys + zs * Y of the source triple
      if (m < 0) continue;
      const int zs = m / p.Y, ys = m - zs * p.Y;
      if (kx == 1) any |= bricks_any(p, vx + p.X, ys, zs);
      if (ys < p.ovy && zs >= 1) any |= bricks_any(p, vx + kx * p.X, ys + p.Y, zs - 1);
    }
  }
  return any;
}

// ---------------------------------------------------------------------------
// Sensors [S0, S0+CNT) of one tile: issue every global load (CNT*3 LUT planes, CNT
// frame windows), one barrier, then fold the 4 voxels of this thread.
// SKIP (RGBDR_FLAG_SKIP_BACKGROUND): what one sensor does to a tile is often known without its LUT planes.  Take a
// tile whose 512 footprints all lie inside the sensor's 16x16 frame window and whose entries are finite, with
// projected depths in [dmin, dmax] (k_tile_windows, at LUT upload), and look at the window's texels
// (k_window_background, once per frame):
//   * all background (silhouette 0), depths <= hi, and fl(dmin - hi) >= limit: the interpolated silhouette is
//     0 + a * (0 - 0) = 0 < 1, so tsdf_integration.vs:34-37 carves -- tsd = -limit where tsd >= limit -- and where it
//     does not, sdist = pc.z - depth >= limit (rounding is monotonic: fl(pc.z - depth) >= fl(dmin - hi)) changes
//     nothing                                                                                   -> kSkipCarve
//   * all surface (silhouette 1: the interpolation gives 1 + a * (1 - 1) = 1, no carve), depths in [lo, hi]:
//       fl(dmax - lo) <= -limit: every voxel lies in front of everything the window shows, sdist <= -limit,
//       tsd = -limit (tsdf_integration.vs:44-45)                                                 -> kSkipFront
//       fl(dmin - hi) >= limit: every voxel is hidden, sdist >= limit, nothing happens           -> kSkipBehind
// k_skip_classify takes the verdicts per tile and frame; a tile with an undecided sensor goes to
// k_integrate_tiled_listed, which applies the verdicts of the others to its four voxels in the sensors' turns and
// leaves their LUT planes (6 KiB per pair) and windows unread.
enum : unsigned { kSkipNone = 0u, kSkipCarve = 1u, kSkipFront = 2u, kSkipBehind = 3u };
template <int CNT, bool NT, bool SKIP>
__device__ __forceinline__ void integrate_group(const IntegrateParams& p, unsigned tile, int q, int s0, int ntot,
                                                uint2 (*win)[kWin * kWinPitch], bool windows_in_use, float limit,
                                                float* tsd, float* wsum, unsigned actions = 0u,
                                                const int* origins = nullptr)
{
  int wx0[CNT], wy0[CNT];
  unsigned act[CNT];
#pragma unroll
  for (int i = 0; i < CNT; ++i) {
    const int d = origins ? origins[s0 + i] : p.win[(size_t)tile * ntot + s0 + i];  // (the listed sweep brings them along)
    wx0[i] = (int)(short)(d & 0xffff);
    wy0[i] = (int)(short)(d >> 16);
    act[i] = SKIP ? (actions >> (2 * (s0 + i))) & 3u : kSkipNone;
  }
  const float4* lut = reinterpret_cast<const float4*>(p.lut_tiled + ((size_t)tile * ntot + s0) * (3 * kTileVoxels)) + q;
  float4 U[CNT], V[CNT], D[CNT];
#pragma unroll
  for (int i = 0; i < CNT; ++i) {
    if (SKIP && act[i] != kSkipNone) continue;
    if (NT) {  // read-once stream: non-temporal, measured 7-8 % faster than default-policy loads
      typedef float v4f __attribute__((ext_vector_type(4)));
      const v4f* l = reinterpret_cast<const v4f*>(lut);
      const v4f u = __builtin_nontemporal_load(&l[(i * 3 + 0) * (kTileVoxels / 4)]);
      const v4f v = __builtin_nontemporal_load(&l[(i * 3 + 1) * (kTileVoxels / 4)]);
      const v4f d = __builtin_nontemporal_load(&l[(i * 3 + 2) * (kTileVoxels / 4)]);
      U[i] = make_float4(u.x, u.y, u.z, u.w);
      V[i] = make_float4(v.x, v.y, v.z, v.w);
      D[i] = make_float4(d.x, d.y, d.z, d.w);
    } else {
      U[i] = lut[(i * 3 + 0) * (kTileVoxels / 4)];
      V[i] = lut[(i * 3 + 1) * (kTileVoxels / 4)];
      D[i] = lut[(i * 3 + 2) * (kTileVoxels / 4)];
    }
  }
  uint2 ta[CNT], tb[CNT];
  const int wr = q >> 3, wc = (q & 7) * 2;
#pragma unroll
  for (int i = 0; i < CNT; ++i) {
    if (SKIP && act[i] != kSkipNone) continue;
    const int row = clampi(wy0[i] + wr, 0, p.H - 1) * p.W;
    ta[i] = p.frame[s0 + i][row + clampi(wx0[i] + wc, 0, p.W - 1)];
    tb[i] = p.frame[s0 + i][row + clampi(wx0[i] + wc + 1, 0, p.W - 1)];
  }
  if (windows_in_use) __syncthreads();  // the previous group's footprints are all read
#pragma unroll
  for (int i = 0; i < CNT; ++i) {
    if (SKIP && act[i] != kSkipNone) continue;
    win[i][wr * kWinPitch + wc] = ta[i];
    win[i][wr * kWinPitch + wc + 1] = tb[i];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < CNT; ++i) {
    if (SKIP && act[i] != kSkipNone) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (act[i] == kSkipCarve) tsd[j] = (tsd[j] >= limit) ? -limit : tsd[j];
        if (act[i] == kSkipFront) tsd[j] = -limit;
      }
      continue;
    }
    const uint2* frame = p.frame[s0 + i];
    fold_voxel_window(win[i], wx0[i], wy0[i], frame, p.W, p.H, U[i].x, V[i].x, D[i].x, limit, tsd[0], wsum[0]);
    fold_voxel_window(win[i], wx0[i], wy0[i], frame, p.W, p.H, U[i].y, V[i].y, D[i].y, limit, tsd[1], wsum[1]);
    fold_voxel_window(win[i], wx0[i], wy0[i], frame, p.W, p.H, U[i].z, V[i].z, D[i].z, limit, tsd[2], wsum[2]);
    fold_voxel_window(win[i], wx0[i], wy0[i], frame, p.W, p.H, U[i].w, V[i].w, D[i].w, limit, tsd[3], wsum[3]);
  }
}

}  // namespace rgbdr
