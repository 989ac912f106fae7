// kernels_invert.hip -- inverse calibration volume on the device.
//
// Replaces the offline CGAL/OpenMP tool CalibrationInverter::calculateInverseVolumes
// (framework/calibration/calibration_inverter.cpp:99-155, source/calib_inverter.cpp):
// for every sample position of the volume, reject it if it lies outside the
// sensor frustum (Frustum::inside, frustum.cpp:36-43), else take the 8 nearest
// cv_xyz samples, weight their LUT indices by inverse distance (:55-69) and store
// (index + 0.5) / dims, 1.
//
// The reference searches a k-d tree over all samples.  The samples form a warped
// regular grid, so this kernel searches locally instead: one cooperative coarse
// scan per 8x8x8 tile finds the sample nearest to the tile centre, every voxel
// walks downhill from there (strides 4, 2, 1) to its own nearest sample, and the 8
// nearest are selected from the (2R+1)^3 index window around it.  Neighbours are
// ordered by (squared distance, sample index) like the oracle's exact search.
//
// EXACTNESS.  A window's result is accepted only with a certificate that no sample
// outside the window can be among the 8 nearest.  The faces of the index window that
// have samples beyond them (its SHELL; faces on the border of the LUT have none) are
// a closed surface of samples between the query q and everything outside, as long as
// the sampled region is convex and the lattice does not fold (a sensor's calibration
// volume is a warped frustum).  A segment from q to an outside sample crosses that
// surface in a facet whose corners are shell samples, and a point of a facet is no
// farther from its nearest corner than the facet's longest edge e, so every outside
// sample is at least  min_shell |q - s| - e  away.  With e bounded by the longest
// lattice edge of the depth slices the window spans (k_lut_edge_max), the window is
// certified when
//        sqrt(d8) < sqrt(min_shell d) - e        (or the window has no shell at all).
// The same inequality fails when the window does not contain q (the walk got stuck):
// then the point of the window nearest to q lies on the shell.  Without a certificate
// the window is re-centred on its nearest sample and widened (R -> R + max(1, R/2), up
// to 8); a voxel still uncertified at R = 8 is appended to a list and k_invert_exhaustive
// scans the WHOLE volume for it, one workgroup per voxel.  So the result equals the exact
// search for every voxel (tests/test_inverter_gpu.py asserts 100 %, not a fraction).
#include <hip/hip_runtime.h>

#include "rgbdr_internal.hpp"

namespace rgbdr {

struct Best8 {
  float d[8];
  int i[8];
};

__device__ __forceinline__ bool nn_less(float d2, int lin, float bd, int bi) { return d2 < bd || (d2 == bd && lin < bi); }

__device__ __forceinline__ float sample_d2(const InvertParams& p, int sx, int sy, int sz, float px, float py, float pz)
{
  const float4 s = p.xyz[((size_t)sz * p.ry + sy) * p.rx + sx];
  const float dx = px - s.x, dy = py - s.y, dz = pz - s.z;
  return dx * dx + dy * dy + dz * dz;
}

__device__ __forceinline__ bool inside_frustum(const InvertParams& p, float px, float py, float pz)
{
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const float d = (p.planes[i][0] * px + p.planes[i][1] * py) + (p.planes[i][2] * pz + p.planes[i][3] * 1.0f);
    if (d < 0.0f) return false;
  }
  return true;
}

// longest lattice edge touching each depth slice, as the bits of its squared length (positive floats order like their bits)
__global__ __launch_bounds__(256) void k_lut_edge_max(const float4* xyz, int rx, int ry, int rz, unsigned* emax2)
{
  const size_t n = (size_t)rx * ry * rz;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int sx = (int)(i % rx), sy = (int)((i / rx) % ry), sz = (int)(i / ((size_t)rx * ry));
  const float4 a = xyz[i];
  auto len2 = [&](size_t j) {
    const float4 b = xyz[j];
    const float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
    return dx * dx + dy * dy + dz * dz;
  };
  float m = 0.0f;
  if (sx + 1 < rx) m = fmaxf(m, len2(i + 1));
  if (sy + 1 < ry) m = fmaxf(m, len2(i + rx));
  if (m > 0.0f) atomicMax(&emax2[sz], __float_as_uint(m));
  if (sz + 1 < rz) {
    const float e = len2(i + (size_t)rx * ry);
    atomicMax(&emax2[sz], __float_as_uint(e));
    atomicMax(&emax2[sz + 1], __float_as_uint(e));
  }
}

// the eight nearest samples -> the record calibration_inverter.cpp:55-69 stores
__device__ __forceinline__ void weigh(const InvertParams& p, const Best8& b, float& u, float& v, float& d)
{
  const int ryz = p.ry * p.rz;
  float tw = 0.0f, wx = 0.0f, wy = 0.0f, wz = 0.0f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (b.i[k] == 0x7fffffff) continue;  // fewer than 8 samples in the LUT
    const float w = 1.0f / sqrtf(b.d[k]);
    wx += w * (float)(b.i[k] / ryz);
    wy += w * (float)((b.i[k] / p.rz) % p.ry);
    wz += w * (float)(b.i[k] % p.rz);
    tw += w;
  }
  u = (wx / tw + 0.5f) / (float)p.rx;
  v = (wy / tw + 0.5f) / (float)p.ry;
  d = (wz / tw + 0.5f) / (float)p.rz;
}

__device__ __forceinline__ void insert8(Best8& b, float d2, int lin)
{
  if (!nn_less(d2, lin, b.d[7], b.i[7])) return;
#pragma unroll
  for (int k = 0; k < 8; ++k) {  // insertion: carry the larger element forward
    if (nn_less(d2, lin, b.d[k], b.i[k])) {
      const float td = b.d[k];
      const int ti = b.i[k];
      b.d[k] = d2;
      b.i[k] = lin;
      d2 = td;
      lin = ti;
    }
  }
}

__global__ __launch_bounds__(128) void k_invert_lut(InvertParams p)
{
  __shared__ float s_d[128];
  __shared__ int s_i[128];
  const unsigned tile = blockIdx.x;
  const int q = threadIdx.x;
  const int lz = q >> 4, ly = (q >> 1) & 7, lx0 = (q & 1) * 4;
  const int tx = tile % p.TX, ty = (tile / p.TX) % p.TY, tzl = tile / (p.TX * p.TY);
  const int vz = p.z0 + tzl * kTile + lz, vy = ty * kTile + ly, vx0 = tx * kTile + lx0;
  const int ryz = p.ry * p.rz;

  // ---- seed: sample nearest to the tile centre, cooperative scan of a stride-4 lattice
  {
    const float cx = p.start[0] + (float)(tx * kTile + 4) * p.step[0];
    const float cy = p.start[1] + (float)(ty * kTile + 4) * p.step[1];
    const float cz = p.start[2] + (float)(p.z0 + tzl * kTile + 4) * p.step[2];
    const int nx = (p.rx + 3) / 4, ny = (p.ry + 3) / 4, nz = (p.rz + 3) / 4;
    float bd = __builtin_inff();
    int bi = 0x7fffffff;
    for (int k = q; k < nx * ny * nz; k += 128) {
      const int sx = min((k % nx) * 4, p.rx - 1), sy = min(((k / nx) % ny) * 4, p.ry - 1), sz = min((k / (nx * ny)) * 4, p.rz - 1);
      const float d2 = sample_d2(p, sx, sy, sz, cx, cy, cz);
      const int lin = (sx * p.ry + sy) * p.rz + sz;
      if (nn_less(d2, lin, bd, bi)) {
        bd = d2;
        bi = lin;
      }
    }
    s_d[q] = bd;
    s_i[q] = bi;
    __syncthreads();
    for (int o = 64; o > 0; o >>= 1) {
      if (q < o && nn_less(s_d[q + o], s_i[q + o], s_d[q], s_i[q])) {
        s_d[q] = s_d[q + o];
        s_i[q] = s_i[q + o];
      }
      __syncthreads();
    }
  }
  const int seed = s_i[0];
  const int seed_x = seed / ryz, seed_y = (seed / p.rz) % p.ry, seed_z = seed % p.rz;

  float ou[4], ov[4], od[4], ow[4];
#pragma unroll 1
  for (int j = 0; j < 4; ++j) {
    const int vx = vx0 + j;
    ou[j] = ov[j] = od[j] = ow[j] = -1.0f;
    if (vx >= p.X || vy >= p.Y || vz >= p.z0 + p.nz) continue;
    const float px = p.start[0] + (float)vx * p.step[0];
    const float py = p.start[1] + (float)vy * p.step[1];
    const float pz = p.start[2] + (float)vz * p.step[2];
    if (!inside_frustum(p, px, py, pz)) continue;
    // ---- downhill walk to the nearest sample
    int cx = seed_x, cy = seed_y, cz = seed_z;
    float cd = sample_d2(p, cx, cy, cz, px, py, pz);
    for (int stride = 4; stride >= 1; stride >>= 1) {
      for (int iter = 0; iter < 256; ++iter) {
        int bx = cx, by = cy, bz = cz;
        float bd = cd;
        for (int dz = -1; dz <= 1; ++dz)
          for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
              const int sx = cx + dx * stride, sy = cy + dy * stride, sz = cz + dz * stride;
              if (sx < 0 || sy < 0 || sz < 0 || sx >= p.rx || sy >= p.ry || sz >= p.rz) continue;
              const float d2 = sample_d2(p, sx, sy, sz, px, py, pz);
              if (d2 < bd) {
                bd = d2;
                bx = sx;
                by = sy;
                bz = sz;
              }
            }
        if (bx == cx && by == cy && bz == cz) break;
        cx = bx;
        cy = by;
        cz = bz;
        cd = bd;
      }
    }
    // ---- 8 nearest of the index window, kept sorted by (d2, sample index); widened until certified
    Best8 b;
    int R = p.window;
    bool certified = false, recentred = false;
#pragma unroll 1
    for (;;) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        b.d[k] = __builtin_inff();
        b.i[k] = 0x7fffffff;
      }
      const int x0 = max(cx - R, 0), x1 = min(cx + R, p.rx - 1);
      const int y0 = max(cy - R, 0), y1 = min(cy + R, p.ry - 1);
      const int z0 = max(cz - R, 0), z1 = min(cz + R, p.rz - 1);
      // faces with samples beyond them
      const bool fx0 = x0 > 0, fx1 = x1 < p.rx - 1, fy0 = y0 > 0, fy1 = y1 < p.ry - 1, fz0 = z0 > 0, fz1 = z1 < p.rz - 1;
      float shell = __builtin_inff();
      for (int sz = z0; sz <= z1; ++sz)
        for (int sy = y0; sy <= y1; ++sy)
          for (int sx = x0; sx <= x1; ++sx) {
            const float d2 = sample_d2(p, sx, sy, sz, px, py, pz);
            const bool on_shell = (fx0 && sx == x0) || (fx1 && sx == x1) || (fy0 && sy == y0) || (fy1 && sy == y1) ||
                                  (fz0 && sz == z0) || (fz1 && sz == z1);
            if (on_shell) shell = fminf(shell, d2);
            insert8(b, d2, (sx * p.ry + sy) * p.rz + sz);
          }
      if (!(fx0 || fx1 || fy0 || fy1 || fz0 || fz1)) {
        certified = true;  // the window is the whole volume
        break;
      }
      unsigned e2 = 0;
      for (int sz = z0; sz <= z1; ++sz) e2 = max(e2, p.emax2[sz]);
      // (rounded against acceptance: the left side up, the right side down by an ulp-scale margin)
      const float lhs = sqrtf(b.d[7]) * 1.000001f, rhs = sqrtf(shell) * 0.999999f - sqrtf(__uint_as_float(e2)) * 1.000001f;
      if (lhs < rhs) {
        certified = true;
        break;
      }
      const int ncx = b.i[0] / ryz, ncy = (b.i[0] / p.rz) % p.ry, ncz = b.i[0] % p.rz;
      if (!recentred && b.i[0] != 0x7fffffff && (ncx != cx || ncy != cy || ncz != cz)) {
        recentred = true;  // the walk stopped short of the nearest sample: same radius around the better centre
        cx = ncx;
        cy = ncy;
        cz = ncz;
        continue;
      }
      if (R >= 8) break;
      R = min(8, R + max(1, R / 2));
      recentred = false;
    }
    if (R != p.window) atomicAdd(&p.stats[0], 1ull);
    if (!certified) {  // exhaustive scan of the whole volume, later (k_invert_exhaustive overwrites this record)
      const unsigned slot = atomicAdd(p.todo_count, 1u);
      p.todo[slot] = (unsigned)(((size_t)(vz - p.z0) * p.Y + vy) * p.X + vx);
    }
    weigh(p, b, ou[j], ov[j], od[j]);
    ow[j] = 1.0f;
  }
  if (p.out_tiled) {
    float4* o = reinterpret_cast<float4*>(p.out_tiled + ((size_t)tile * p.N + p.sensor) * 3 * kTileVoxels) + q;
    o[0] = make_float4(ou[0], ou[1], ou[2], ou[3]);
    o[kTileVoxels / 4] = make_float4(ov[0], ov[1], ov[2], ov[3]);
    o[2 * (kTileVoxels / 4)] = make_float4(od[0], od[1], od[2], od[3]);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int vx = vx0 + j;
      if (vx >= p.X || vy >= p.Y || vz >= p.z0 + p.nz) continue;
      p.out_linear[((size_t)(vz - p.z0) * p.Y + vy) * p.X + vx] = make_float4(ou[j], ov[j], od[j], ow[j]);
    }
  }
}

// One workgroup per uncertified voxel: every sample of the volume, the 256 partial lists merged pairwise in LDS.
__global__ __launch_bounds__(256) void k_invert_exhaustive(InvertParams p)
{
  __shared__ float s_d[256][8];
  __shared__ int s_i[256][8];
  const unsigned count = *p.todo_count;
  const int q = threadIdx.x;
  const size_t n = (size_t)p.rx * p.ry * p.rz;
  for (unsigned e = blockIdx.x; e < count; e += gridDim.x) {
    const unsigned v = p.todo[e];
    const int vx = (int)(v % (unsigned)p.X), vy = (int)((v / (unsigned)p.X) % (unsigned)p.Y), lz = (int)(v / ((unsigned)p.X * p.Y));
    const float px = p.start[0] + (float)vx * p.step[0];
    const float py = p.start[1] + (float)vy * p.step[1];
    const float pz = p.start[2] + (float)(p.z0 + lz) * p.step[2];
    Best8 b;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      b.d[k] = __builtin_inff();
      b.i[k] = 0x7fffffff;
    }
    for (size_t i = q; i < n; i += 256) {
      const int sx = (int)(i % p.rx), sy = (int)((i / p.rx) % p.ry), sz = (int)(i / ((size_t)p.rx * p.ry));
      insert8(b, sample_d2(p, sx, sy, sz, px, py, pz), (sx * p.ry + sy) * p.rz + sz);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      s_d[q][k] = b.d[k];
      s_i[q][k] = b.i[k];
    }
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (q < o) {
        for (int k = 0; k < 8; ++k) insert8(b, s_d[q + o][k], s_i[q + o][k]);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          s_d[q][k] = b.d[k];
          s_i[q][k] = b.i[k];
        }
      }
      __syncthreads();
    }
    if (q == 0) {
      float u, w, d;
      weigh(p, b, u, w, d);
      if (p.out_tiled) {
        const int tile = ((lz / kTile) * p.TY + vy / kTile) * p.TX + vx / kTile;
        const int local = ((lz % kTile) * kTile + vy % kTile) * kTile + vx % kTile;
        float* o = p.out_tiled + ((size_t)tile * p.N + p.sensor) * 3 * kTileVoxels + local;
        o[0] = u;
        o[kTileVoxels] = w;
        o[2 * kTileVoxels] = d;
      } else {
        p.out_linear[v] = make_float4(u, w, d, 1.0f);
      }
      atomicAdd(&p.stats[1], 1ull);
    }
    __syncthreads();
  }
}

void launch_lut_edge_max(const float4* xyz, int rx, int ry, int rz, unsigned* emax2, hipStream_t s)
{
  const size_t n = (size_t)rx * ry * rz;
  hipLaunchKernelGGL(k_lut_edge_max, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, xyz, rx, ry, rz, emax2);
}

void launch_invert_exhaustive(const InvertParams& p, hipStream_t s)
{
  hipLaunchKernelGGL(k_invert_exhaustive, dim3(2048), dim3(256), 0, s, p);
}

void launch_invert_lut(const InvertParams& p, hipStream_t s)
{
  const unsigned tz = (unsigned)((p.nz + kTile - 1) / kTile);
  hipLaunchKernelGGL(k_invert_lut, dim3((unsigned)p.TX * p.TY * tz), dim3(128), 0, s, p);
}

}  // namespace rgbdr
