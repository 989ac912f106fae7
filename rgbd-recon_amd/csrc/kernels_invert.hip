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
// a closed surface between the query q and everything outside, as long as the sampled
// region is convex and the lattice does not fold (a sensor's calibration volume is a
// warped frustum): a segment from q to an outside sample crosses that surface in a
// patch of one face, and a (bi)linear patch lies in the convex hull of its corner
// samples.  So for ANY unit vector u and any point p of face f
//        |p - q|  >=  u . (p - q)  >=  min over the samples s of f of  u . (s - q),
// and with u_f = the outward normal of the lattice at the window's centre (cross product
// of the two in-face index directions) the right side is the distance of q to the face
// when the face is flat -- no slack for the size or the elongation of the cells.  The
// window is certified when
//        sqrt(d8)  <  min over its shell faces f of  min_s u_f . (s - q)
// (or it has no shell at all).  The inequality also fails when the window does not
// contain q (the walk got stuck): the face towards q then has samples behind q.  Without
// a certificate the window is re-centred on its nearest sample and widened (R -> R +
// max(1, R/2), up to 8); a voxel still uncertified at R = 8 is appended to a list and
// k_invert_exhaustive scans the WHOLE volume for it, one workgroup per voxel.  So the
// result equals the exact search for every voxel (tests/test_inverter_gpu.py asserts
// equality, not a fraction).  The argument needs a lattice whose cells all have one orientation; the host checks that when
// the calibration is set (geometry.cpp lattice_folds) and, where a cv_xyz volume folds, p.folded sends every voxel to the
// exhaustive scan (a radial distortion of k = -0.3 folds near the corners: 6 of 168 000 voxels were wrong with a
// "certificate" there, tests/test_inverter_gpu.py).
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

__device__ __forceinline__ void write_record(const InvertParams& p, int vx, int vy, int lz, unsigned v, float u, float w, float d)
{
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

// n = normalize(cross(a, b)), oriented along `towards`
__device__ __forceinline__ void unit_normal(const float* a, const float* b, const float* towards, float* n)
{
  const float cx = a[1] * b[2] - a[2] * b[1], cy = a[2] * b[0] - a[0] * b[2], cz = a[0] * b[1] - a[1] * b[0];
  const float len = sqrtf(cx * cx + cy * cy + cz * cz);
  const float sgn = (cx * towards[0] + cy * towards[1] + cz * towards[2]) < 0.0f ? -1.0f : 1.0f;
  n[0] = sgn * cx / len;
  n[1] = sgn * cy / len;
  n[2] = sgn * cz / len;
}

// the lattice's unit normals at sample (cx, cy, cz): of the faces of constant x (pointing towards +x), y, z -- cross products of
// the two in-face index directions (central differences, one-sided at the border)
__device__ __forceinline__ void lattice_normals(const InvertParams& p, int cx, int cy, int cz, float* nx, float* ny, float* nz)
{
  const float4 xa = p.xyz[((size_t)cz * p.ry + cy) * p.rx + max(cx - 1, 0)], xb = p.xyz[((size_t)cz * p.ry + cy) * p.rx + min(cx + 1, p.rx - 1)];
  const float4 ya = p.xyz[((size_t)cz * p.ry + max(cy - 1, 0)) * p.rx + cx], yb = p.xyz[((size_t)cz * p.ry + min(cy + 1, p.ry - 1)) * p.rx + cx];
  const float4 za = p.xyz[((size_t)max(cz - 1, 0) * p.ry + cy) * p.rx + cx], zb = p.xyz[((size_t)min(cz + 1, p.rz - 1) * p.ry + cy) * p.rx + cx];
  const float ex[3] = {xb.x - xa.x, xb.y - xa.y, xb.z - xa.z};
  const float ey[3] = {yb.x - ya.x, yb.y - ya.y, yb.z - ya.z};
  const float ez[3] = {zb.x - za.x, zb.y - za.y, zb.z - za.z};
  unit_normal(ey, ez, ex, nx);
  unit_normal(ez, ex, ey, ny);
  unit_normal(ex, ey, ez, nz);
}

// R0: the radius of the first window as a compile-time constant (1, 2, 3: its loops unroll and the loads of a row of
// samples are in flight together) or 0 = take p.window at run time
template <int R0>
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
  int cx = seed_x, cy = seed_y, cz = seed_z;
  bool warm = false;  // (cx, cy, cz) is the nearest sample of the lane's previous voxel, 1 / X of the box away
#pragma unroll 1
  for (int j = 0; j < 4; ++j) {
    const int vx = vx0 + j;
    ou[j] = ov[j] = od[j] = ow[j] = -1.0f;
    if (vx >= p.X || vy >= p.Y || vz >= p.z0 + p.nz) continue;
    const float px = p.start[0] + (float)vx * p.step[0];
    const float py = p.start[1] + (float)vy * p.step[1];
    const float pz = p.start[2] + (float)vz * p.step[2];
    if (!inside_frustum(p, px, py, pz)) continue;
    // ---- downhill walk to the nearest sample: from the tile's seed with strides 4, 2, 1, or from the neighbouring voxel's
    // result with stride 1 (wherever it ends, the certificate below decides whether the window is accepted)
    if (!warm) {
      cx = seed_x;
      cy = seed_y;
      cz = seed_z;
    }
    float cd = sample_d2(p, cx, cy, cz, px, py, pz);
    for (int stride = warm ? 1 : 4; stride >= 1; stride >>= 1) {
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
    warm = true;
    const int wcx = cx, wcy = cy, wcz = cz;  // where the walk ended (the next voxel starts here, whatever happens below)
    // ---- 8 nearest of the index window, kept sorted by (d2, sample index)
    Best8 b;
    const int R = R0 > 0 ? R0 : p.window;
    bool certified = false;
    {
      // The FIRST window: the same (2R+1)^3 offsets for every lane of the wavefront, so the loops are scalar loops, the
      // "is this sample on face f" tests are wave-uniform, and only the outward distances of the faces an offset lies on are
      // computed.  Offsets that fall outside the volume are skipped (that IS the clipped window of the general case below).
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        b.d[k] = __builtin_inff();
        b.i[k] = 0x7fffffff;
      }
      const bool fx0 = cx - R > 0, fx1 = cx + R < p.rx - 1, fy0 = cy - R > 0, fy1 = cy + R < p.ry - 1, fz0 = cz - R > 0, fz1 = cz + R < p.rz - 1;
      const bool any_shell = fx0 || fx1 || fy0 || fy1 || fz0 || fz1;
      float nx[3] = {0.0f, 0.0f, 0.0f}, ny[3] = {0.0f, 0.0f, 0.0f}, nz[3] = {0.0f, 0.0f, 0.0f};
      if (any_shell) lattice_normals(p, cx, cy, cz, nx, ny, nz);
      float plane = __builtin_inff();  // smallest outward distance of a shell sample along its face's normal
      if (R0 > 0) {
        // a row of the window at a time: its 2 R + 1 loads are issued together (clamped addresses, so none is conditional),
        // then the samples that lie inside the volume are looked at
        constexpr int ROW = R0 > 0 ? 2 * R0 + 1 : 1;
#pragma unroll 1
        for (int oz = -R; oz <= R; ++oz) {
          const int sz = cz + oz;
          const bool zin = (unsigned)sz < (unsigned)p.rz;
#pragma unroll 1
          for (int oy = -R; oy <= R; ++oy) {
            const int sy = cy + oy;
            const bool rin = zin && (unsigned)sy < (unsigned)p.ry;
            const float4* row = p.xyz + ((size_t)min(max(sz, 0), p.rz - 1) * p.ry + min(max(sy, 0), p.ry - 1)) * p.rx;
            float4 sp[ROW];
#pragma unroll
            for (int k = 0; k < ROW; ++k) sp[k] = row[min(max(cx + k - R, 0), p.rx - 1)];
            // the outward distances of the faces this ROW lies on (wave-uniform tests)
            const bool on_y = oy == -R || oy == R, on_z = oz == -R || oz == R;
            const bool use_y = on_y && (oy == -R ? fy0 : fy1), use_z = on_z && (oz == -R ? fz0 : fz1);
#pragma unroll
            for (int k = 0; k < ROW; ++k) {
              const int ox = k - R, sx = cx + ox;
              if (!rin || (unsigned)sx >= (unsigned)p.rx) continue;
              const float dx = px - sp[k].x, dy = py - sp[k].y, dz = pz - sp[k].z;
              const float d2 = dx * dx + dy * dy + dz * dz;
              if (ox == -R || ox == R) {  // compile-time
                const float t = nx[0] * dx + nx[1] * dy + nx[2] * dz;  // n . (q - s): the outward distance of a LOW face
                if (ox == -R ? fx0 : fx1) plane = fminf(plane, ox == -R ? t : -t);
              }
              if (on_y) {
                const float t = ny[0] * dx + ny[1] * dy + ny[2] * dz;
                if (use_y) plane = fminf(plane, oy == -R ? t : -t);
              }
              if (on_z) {
                const float t = nz[0] * dx + nz[1] * dy + nz[2] * dz;
                if (use_z) plane = fminf(plane, oz == -R ? t : -t);
              }
              insert8(b, d2, (sx * p.ry + sy) * p.rz + sz);
            }
          }
        }
      } else {
        for (int oz = -R; oz <= R; ++oz)
          for (int oy = -R; oy <= R; ++oy)
            for (int ox = -R; ox <= R; ++ox) {
              const int sx = cx + ox, sy = cy + oy, sz = cz + oz;
              if ((unsigned)sx >= (unsigned)p.rx || (unsigned)sy >= (unsigned)p.ry || (unsigned)sz >= (unsigned)p.rz) continue;
              const float4 sp = p.xyz[((size_t)sz * p.ry + sy) * p.rx + sx];
              const float dx = px - sp.x, dy = py - sp.y, dz = pz - sp.z;
              const float d2 = dx * dx + dy * dy + dz * dz;
              if (ox == -R || ox == R) {  // wave-uniform
                const float t = nx[0] * dx + nx[1] * dy + nx[2] * dz;  // n . (q - s): the outward distance of a LOW face
                if (ox == -R ? fx0 : fx1) plane = fminf(plane, ox == -R ? t : -t);
              }
              if (oy == -R || oy == R) {
                const float t = ny[0] * dx + ny[1] * dy + ny[2] * dz;
                if (oy == -R ? fy0 : fy1) plane = fminf(plane, oy == -R ? t : -t);
              }
              if (oz == -R || oz == R) {
                const float t = nz[0] * dx + nz[1] * dy + nz[2] * dz;
                if (oz == -R ? fz0 : fz1) plane = fminf(plane, oz == -R ? t : -t);
              }
              insert8(b, d2, (sx * p.ry + sy) * p.rz + sz);
            }
      }
      // (rounded against acceptance; a NaN normal -- a degenerate lattice -- compares false: not certified)
      certified = !any_shell || (!p.folded && plane > 0.0f && sqrtf(b.d[7]) * 1.00001f < plane * 0.99999f);
    }
    if (!certified) {  // the general case (re-centring, widening) is k_invert_retry's: a second, rarely needed kernel
      const unsigned slot = atomicAdd(p.retry_count, 1u);
      p.retry[2 * slot] = (unsigned)(((size_t)(vz - p.z0) * p.Y + vy) * p.X + vx);
      p.retry[2 * slot + 1] = (unsigned)((cx * p.ry + cy) * p.rz + cz);
    }
    cx = wcx;
    cy = wcy;
    cz = wcz;
    weigh(p, b, ou[j], ov[j], od[j]);  // (provisional where the window was not certified: k_invert_retry overwrites it)
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

// The voxels whose first window was not certified (k_invert_lut's list: voxel, the sample its walk ended at): the general
// search -- per-lane window bounds, re-centring on the window's nearest sample, widening up to R = 8 -- one thread per
// voxel.  What is still uncertified then goes on the list of k_invert_exhaustive.
__global__ __launch_bounds__(64) void k_invert_retry(InvertParams p)
{
  const unsigned count = *p.retry_count;
  const int ryz = p.ry * p.rz;
  for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < count; e += gridDim.x * blockDim.x) {
    const unsigned v = p.retry[2 * e], c = p.retry[2 * e + 1];
    const int vx = (int)(v % (unsigned)p.X), vy = (int)((v / (unsigned)p.X) % (unsigned)p.Y), lz = (int)(v / ((unsigned)p.X * p.Y));
    const float px = p.start[0] + (float)vx * p.step[0];
    const float py = p.start[1] + (float)vy * p.step[1];
    const float pz = p.start[2] + (float)(p.z0 + lz) * p.step[2];
    int cx = (int)c / ryz, cy = ((int)c / p.rz) % p.ry, cz = (int)c % p.rz;
    Best8 b;
    int R = p.window;
    bool certified = false, recentred = false;
#pragma unroll 1
    while (!certified) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        b.d[k] = __builtin_inff();
        b.i[k] = 0x7fffffff;
      }
      const int x0 = max(cx - R, 0), x1 = min(cx + R, p.rx - 1);
      const int y0 = max(cy - R, 0), y1 = min(cy + R, p.ry - 1);
      const int z0 = max(cz - R, 0), z1 = min(cz + R, p.rz - 1);
      // faces with samples beyond them, and the lattice's outward directions at the centre
      const bool fx0 = x0 > 0, fx1 = x1 < p.rx - 1, fy0 = y0 > 0, fy1 = y1 < p.ry - 1, fz0 = z0 > 0, fz1 = z1 < p.rz - 1;
      const bool any_shell = fx0 || fx1 || fy0 || fy1 || fz0 || fz1;
      float nx[3], ny[3], nz[3];
      if (any_shell) lattice_normals(p, cx, cy, cz, nx, ny, nz);
      float plane = __builtin_inff();
      for (int sz = z0; sz <= z1; ++sz)
        for (int sy = y0; sy <= y1; ++sy)
          for (int sx = x0; sx <= x1; ++sx) {
            const float4 sp = p.xyz[((size_t)sz * p.ry + sy) * p.rx + sx];
            const float dx = px - sp.x, dy = py - sp.y, dz = pz - sp.z;
            const float d2 = dx * dx + dy * dy + dz * dz;
            if (any_shell) {
              const float tx = -(nx[0] * dx + nx[1] * dy + nx[2] * dz);  // n . (s - q)
              const float ty = -(ny[0] * dx + ny[1] * dy + ny[2] * dz);
              const float tz = -(nz[0] * dx + nz[1] * dy + nz[2] * dz);
              if (fx0 && sx == x0) plane = fminf(plane, -tx);
              if (fx1 && sx == x1) plane = fminf(plane, tx);
              if (fy0 && sy == y0) plane = fminf(plane, -ty);
              if (fy1 && sy == y1) plane = fminf(plane, ty);
              if (fz0 && sz == z0) plane = fminf(plane, -tz);
              if (fz1 && sz == z1) plane = fminf(plane, tz);
            }
            insert8(b, d2, (sx * p.ry + sy) * p.rz + sz);
          }
      if (!any_shell) {
        certified = true;  // the window is the whole volume
        break;
      }
      if (!p.folded && plane > 0.0f && sqrtf(b.d[7]) * 1.00001f < plane * 0.99999f) {
        certified = true;
        break;
      }
      if (p.folded) break;  // no certificate holds on a lattice that folds: straight to the exhaustive scan
      const int ncx = b.i[0] / ryz, ncy = (b.i[0] / p.rz) % p.ry, ncz = b.i[0] % p.rz;
      if (!recentred && b.i[0] != 0x7fffffff && (ncx != cx || ncy != cy || ncz != cz)) {
        recentred = true;
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
    if (!certified) {
      const unsigned slot = atomicAdd(p.todo_count, 1u);
      p.todo[slot] = v;
    }
    float u, w, d;
    weigh(p, b, u, w, d);
    write_record(p, vx, vy, lz, v, u, w, d);
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
      write_record(p, vx, vy, lz, v, u, w, d);
      atomicAdd(&p.stats[1], 1ull);
    }
    __syncthreads();
  }
}

void launch_invert_retry(const InvertParams& p, hipStream_t s)
{
  hipLaunchKernelGGL(k_invert_retry, dim3(4096), dim3(64), 0, s, p);
}

void launch_invert_exhaustive(const InvertParams& p, hipStream_t s)
{
  hipLaunchKernelGGL(k_invert_exhaustive, dim3(2048), dim3(256), 0, s, p);
}

void launch_invert_lut(const InvertParams& p, hipStream_t s)
{
  const unsigned tz = (unsigned)((p.nz + kTile - 1) / kTile);
  const dim3 grid((unsigned)p.TX * p.TY * tz);
  switch (p.window) {
    case 1: hipLaunchKernelGGL(k_invert_lut<1>, grid, dim3(128), 0, s, p); break;
    case 2: hipLaunchKernelGGL(k_invert_lut<2>, grid, dim3(128), 0, s, p); break;
    case 3: hipLaunchKernelGGL(k_invert_lut<3>, grid, dim3(128), 0, s, p); break;
    default: hipLaunchKernelGGL(k_invert_lut<0>, grid, dim3(128), 0, s, p); break;
  }
}

}  // namespace rgbdr
