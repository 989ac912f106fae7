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
// ordered by (squared distance, sample index) like the oracle's exact search, so
// wherever the window holds the true 8 nearest the result is bit-identical;
// DESIGN.md states how often that is and the reprojection error otherwise.
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
    // ---- 8 nearest of the index window, kept sorted by (d2, sample index)
    Best8 b;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      b.d[k] = __builtin_inff();
      b.i[k] = 0x7fffffff;
    }
    const int R = p.window;
    const int x0 = max(cx - R, 0), x1 = min(cx + R, p.rx - 1);
    const int y0 = max(cy - R, 0), y1 = min(cy + R, p.ry - 1);
    const int z0 = max(cz - R, 0), z1 = min(cz + R, p.rz - 1);
    for (int sz = z0; sz <= z1; ++sz)
      for (int sy = y0; sy <= y1; ++sy)
        for (int sx = x0; sx <= x1; ++sx) {
          float d2 = sample_d2(p, sx, sy, sz, px, py, pz);
          int lin = (sx * p.ry + sy) * p.rz + sz;
          if (!nn_less(d2, lin, b.d[7], b.i[7])) continue;
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
    ou[j] = (wx / tw + 0.5f) / (float)p.rx;
    ov[j] = (wy / tw + 0.5f) / (float)p.ry;
    od[j] = (wz / tw + 0.5f) / (float)p.rz;
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

void launch_invert_lut(const InvertParams& p, hipStream_t s)
{
  const unsigned tz = (unsigned)((p.nz + kTile - 1) / kTile);
  hipLaunchKernelGGL(k_invert_lut, dim3((unsigned)p.TX * p.TY * tz), dim3(128), 0, s, p);
}

}  // namespace rgbdr
