// kernels_integrate.hip -- TSDF integration for gfx950 (MI355X), replacing
// glsl/tsdf_integration.vs driven by ReconIntegration::integrate
// (framework/reconstruction/recon_integration.cpp:243-270) and the voxel-centre
// VBO of VolumeSampler (framework/rendering/volume_sampler.cpp:33-76).
//
// Layout (DESIGN.md "Data layout in HBM"):
//   * TSDF volume: tile-linear, 8x8x8-voxel tiles of 2 KiB, tiles x-fastest.
//   * 1:1 inverse LUT: per tile, per sensor, three 512-float planes (u, v, d) --
//     the .w channel of the reference's RGBA32F texels is never read by the
//     shader (tsdf_integration.vs:31 takes .xyz) and is dropped at upload.
//     One workgroup (2 wavefronts) sweeps one tile: every LUT plane is a
//     contiguous 2 KiB stream read with 16-B-per-lane loads, the 2 KiB tile of
//     TSDF leaves with 16-B-per-lane stores, and the clear to -limit is fused
//     (no separate memset pass, recon_integration.cpp:249-251).
//   * generic inverse LUT (resolution != TSDF resolution): the file's x-fastest
//     RGBA32F volume, sampled with 8 taps per voxel.
//   * frames: per sensor H*W 8-B texels {depth_b.r, quality | !silhouette << 31}:
//     one texel serves all three samplers of the shader (tsdf_integration.vs:32,40,50).
//
// The per-voxel fold over sensors is order dependent (overwrite-to -limit
// branches interleaved with a running weighted mean, tsdf_integration.vs:28-55),
// so sensors stay sequential per lane and voxels are parallel across lanes.
// Bound: HBM streaming, (4 + 12 N) B per voxel at 1:1.  No MFMA: there is no
// dense contraction anywhere on this path.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "integrate_fold.cuh"

namespace rgbdr {

// 1:1 LUT.  128 threads (2 wavefronts) sweep one 8x8x8 tile; thread q owns voxels
// x0..x0+3 of row (y,z).  Every global load of a sensor group -- the 3 LUT planes
// per sensor (16 B per lane, fully coalesced) and the 16x16 frame window per sensor
// the tile projects into (origins precomputed per tile at LUT upload) -- is issued
// before that group's barrier, so one memory latency is paid per group; the 2x2
// footprints of all 512 voxels are then served from LDS.  More than 4 sensors are
// folded in two groups, eight in three (the running tsd / weight stay in registers), which keeps
// the kernel at <= 96 VGPRs = five wavefronts per SIMD for every N.
// One tile: BRICKS marks the voxels of unoccupied bricks -limit after the fold.
// Sensors per group: every sensor in one group up to MAXG, two halves above that -- except for eight sensors, which are
// folded as (3, 3, 2): 36 instead of 48 registers of LUT planes in flight take that kernel from 104 VGPRs (four
// wavefronts per SIMD) to 94 (five, like the 4-sensor kernel); measured 2.09 -> 2.05 ms at 512^3 (round 4; groups of two
// the same, seven sensors as (3, 3, 1) no better than (4, 3)).
template <int N, int MAXG>
struct Groups {
  static constexpr bool kThrees = N == 8;
  static constexpr int G1 = kThrees ? 3 : (N <= MAXG ? N : (N + 1) / 2);
  static constexpr int G2 = kThrees ? 3 : N - G1;
  static constexpr int G3 = N - G1 - G2;
};

template <int N, int MAXG, bool NT, bool ELIDE = false, bool STAGE = false>
__device__ __forceinline__ void integrate_tile(const IntegrateParams& p, unsigned tile, uint2 (*win)[kWin * kWinPitch])
{
  typedef Groups<N, MAXG> G;
  constexpr int G1 = G::G1, G2 = G::G2, G3 = G::G3;
  const int q = threadIdx.x;
  float4* out = reinterpret_cast<float4*>(p.tsdf + (size_t)tile * kTileVoxels) + q;
  const float limit = p.limit;
  float tsd[4] = {limit, limit, limit, limit};
  float wsum[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  integrate_group<G1, NT, false>(p, tile, q, 0, N, win, false, limit, tsd, wsum);
  if (G2 > 0) integrate_group<(G2 > 0 ? G2 : 1), NT, false>(p, tile, q, G1, N, win, true, limit, tsd, wsum);
  if (G3 > 0) integrate_group<(G3 > 0 ? G3 : 1), NT, false>(p, tile, q, G1 + G2, N, win, true, limit, tsd, wsum);
  if (ELIDE) {
    // RGBDR_FLAG_ELIDE_STORES: a tile that comes out all -limit and has held -limit since a sweep
    // of this epoch (tile_state, see k_brick_clear) need not be written again
    const float ml = -limit;
    const bool mine = tsd[0] == ml && tsd[1] == ml && tsd[2] == ml && tsd[3] == ml;
    const unsigned st = p.tile_state[tile];  // read before the barrier: lane 0 rewrites it after
    const bool all_clear = __syncthreads_and(mine) != 0;
    if (all_clear && st == p.epoch) return;
    if (q == 0) p.tile_state[tile] = all_clear ? p.epoch : 0u;
  }
  if (NT) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    v4f r = {tsd[0], tsd[1], tsd[2], tsd[3]};
    __builtin_nontemporal_store(r, reinterpret_cast<v4f*>(out));
  } else
    *out = make_float4(tsd[0], tsd[1], tsd[2], tsd[3]);
  if (STAGE) {  // boundary tile layers of a Z slab go to the halo staging buffers as well (write-once streams too: non-temporal)
    typedef float v4f __attribute__((ext_vector_type(4)));
    const unsigned per_layer = (unsigned)(p.TX * p.TY), layer = tile / per_layer;
    const v4f r4 = {tsd[0], tsd[1], tsd[2], tsd[3]};
    if (p.stage_lo && layer < (unsigned)p.stage_layers)
      __builtin_nontemporal_store(r4, reinterpret_cast<v4f*>(p.stage_lo + (size_t)tile * kTileVoxels) + q);
    const unsigned first_hi = (unsigned)(p.ntz - p.stage_layers);
    if (p.stage_hi && layer >= first_hi)
      __builtin_nontemporal_store(r4, reinterpret_cast<v4f*>(p.stage_hi + (size_t)(tile - first_hi * per_layer) * kTileVoxels) + q);
  }
}

// Full sweep: one block per tile.
template <int N, int MAXG = 4, bool NT = true, bool ELIDE = false, bool STAGE = false>
__global__ __launch_bounds__(128) void k_integrate_tiled(IntegrateParams p)
{
  constexpr int G1 = Groups<N, MAXG>::G1;
  __shared__ uint2 win[G1][kWin * kWinPitch];
  // XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch).
  // Each XCD takes chunks of `order_chunk` consecutive tiles (neighbouring tiles
  // project into overlapping frame windows -> hits in that XCD's L2), and the 8
  // XCDs work on 8 adjacent chunks at a time so the concurrent LUT streams stay
  // close together in memory.
  // (block_base is a multiple of 8: block b of a later launch lands on XCD b & 7 like block b of a single one)
  const unsigned b = blockIdx.x + p.block_base;
  unsigned tile = b;
  if (p.order_chunk) {
    const unsigned xcd = b & 7u, idx = b >> 3;
    const unsigned chunk = idx / p.order_chunk, within = idx - chunk * p.order_chunk;
    tile = (chunk * 8u + xcd) * p.order_chunk + within;
  }
  integrate_tile<N, MAXG, NT, ELIDE, STAGE>(p, tile, win);
}

// Window origin of one (tile, sensor): the minimum footprint index over the
// tile's voxels whose footprint lies inside the image (others -- invalid -1
// entries, far-off projections -- take the global path in the kernel).
__global__ __launch_bounds__(128) void k_tile_windows(const float* __restrict__ lut_tiled, int W, int H, int sensor,
                                                      int N, int32_t* __restrict__ win, float* __restrict__ win_dmin,
                                                      float* __restrict__ win_dmax, int32_t* __restrict__ win_ext)
{
  __shared__ int smin[2][2];
  __shared__ int org[2];
  __shared__ float sdmin[2], sdmax[2];
  __shared__ int sext[2];
  const unsigned tile = blockIdx.x;
  const int q = threadIdx.x;
  const float4* lut = reinterpret_cast<const float4*>(lut_tiled + ((size_t)tile * N + sensor) * 3 * kTileVoxels) + q;
  const float4 U = lut[0], V = lut[kTileVoxels / 4];
  const float us[4] = {U.x, U.y, U.z, U.w}, vs[4] = {V.x, V.y, V.z, V.w};
  int mx = 0x7fffffff, my = 0x7fffffff, ix = 0x7fffffff, iy = 0x7fffffff;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float a;
    const int jx = footprint(us[j], W, a), jy = footprint(vs[j], H, a);
    if (jx >= -1 && jx <= W - 1 && jy >= -1 && jy <= H - 1) {
      // entries carrying the inverter's "outside the frustum" marker (-1,-1) all hit
      // texel (0,0) through a broadcast global fetch; keep them out of the window
      // placement unless the whole tile is such (then the window sits at (-1,-1))
      if (us[j] == -1.0f && vs[j] == -1.0f) {
        ix = min(ix, jx);
        iy = min(iy, jy);
      } else {
        mx = min(mx, jx);
        my = min(my, jy);
      }
    }
  }
  if (__syncthreads_and(mx == 0x7fffffff)) {  // no valid footprint in the tile
    mx = ix;
    my = iy;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mx = min(mx, __shfl_xor(mx, o));
    my = min(my, __shfl_xor(my, o));
  }
  if ((q & 63) == 0) {
    smin[q >> 6][0] = mx;
    smin[q >> 6][1] = my;
  }
  __syncthreads();
  if (q == 0) {
    mx = min(smin[0][0], smin[1][0]);
    my = min(smin[0][1], smin[1][1]);
    if (mx == 0x7fffffff) mx = my = 0;
    win[(size_t)tile * N + sensor] = (int32_t)(((uint32_t)(my & 0xffff) << 16) | (uint32_t)(mx & 0xffff));
    org[0] = (int)(short)(mx & 0xffff);  // as the sweep decodes it
    org[1] = (int)(short)(my & 0xffff);
  }
  __syncthreads();
  // RGBDR_FLAG_SKIP_BACKGROUND: the smallest and largest projected depth of the tile, provided every entry is finite
  // and every 2x2 footprint lies inside the window (then the sweep touches no texel outside it); -inf / +inf otherwise
  const float4 D = lut[2 * (kTileVoxels / 4)];
  const float ds[4] = {D.x, D.y, D.z, D.w};
  bool ok = true;
  float dm = __builtin_inff(), dx = -__builtin_inff();
  int ext = 0;  // texels per axis the footprints span from the window origin
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float a;
    const int rx = footprint(us[j], W, a) - org[0], ry = footprint(vs[j], H, a) - org[1];
    ok = ok && (unsigned)rx < (unsigned)(kWin - 1) && (unsigned)ry < (unsigned)(kWin - 1) &&
         fabsf(us[j]) < __builtin_inff() && fabsf(vs[j]) < __builtin_inff() && fabsf(ds[j]) < __builtin_inff();
    dm = fminf(dm, ds[j]);
    dx = fmaxf(dx, ds[j]);
    ext = max(ext, max(rx, ry) + 2);
  }
  const bool all_ok = __syncthreads_and(ok) != 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    dm = fminf(dm, __shfl_xor(dm, o));
    dx = fmaxf(dx, __shfl_xor(dx, o));
    ext = max(ext, __shfl_xor(ext, o));
  }
  if ((q & 63) == 0) {
    sdmin[q >> 6] = dm;
    sdmax[q >> 6] = dx;
    sext[q >> 6] = ext;
  }
  __syncthreads();
  if (q == 0) {
    win_dmin[(size_t)tile * N + sensor] = all_ok ? fminf(sdmin[0], sdmin[1]) : -__builtin_inff();
    win_dmax[(size_t)tile * N + sensor] = all_ok ? fmaxf(sdmax[0], sdmax[1]) : __builtin_inff();
    // the square of 4, 8 or 16 texels at the window origin that holds every footprint (k_window_background has
    // bounds for each size: the smaller the square, the more often it is all background or all surface)
    const int e = max(sext[0], sext[1]);
    win_ext[(size_t)tile * N + sensor] = e <= 4 ? 0 : (e <= 8 ? 1 : 2);
  }
}

// `win` holds four planes of ntiles * N words: the window origins, the tiles' smallest and largest projected depths,
// the size class of the footprints' square
void launch_tile_windows(const float* lut_tiled, int W, int H, int ntiles, int sensor, int N, int32_t* win,
                         hipStream_t s)
{
  hipLaunchKernelGGL(k_tile_windows, dim3((unsigned)ntiles), dim3(128), 0, s, lut_tiled, W, H, sensor, N, win,
                     reinterpret_cast<float*>(win + (size_t)ntiles * N), reinterpret_cast<float*>(win + 2 * (size_t)ntiles * N),
                     win + 3 * (size_t)ntiles * N);
}

// ---------------------------------------------------------------------------
// Generic LUT resolution: 8-tap trilinear of the RGBA32F volume per voxel, frame
// footprints gathered from global memory.
template <bool BRICKS>
__global__ __launch_bounds__(128) void k_integrate_generic(IntegrateParams p)
{
  const unsigned tile = blockIdx.x;
  const int q = threadIdx.x;
  const int lz = q >> 4, ly = (q >> 1) & 7, lx0 = (q & 1) * 4;
  const int tx = tile % p.TX;
  const int ty = (tile / p.TX) % p.TY;
  const int tzl = tile / (p.TX * p.TY);
  const int vz = (p.tz0 + tzl) * kTile + lz, vy = ty * kTile + ly, vx0 = tx * kTile + lx0;
  float4* out = reinterpret_cast<float4*>(p.tsdf + (size_t)tile * kTileVoxels) + q;
  const float limit = p.limit;
  float res[4];
  // voxel centre exactly as VolumeSampler builds it (volume_sampler.cpp:36-42)
  const float pz = ((float)vz + 0.5f) * p.stepZ, py = ((float)vy + 0.5f) * p.stepY;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int vx = vx0 + j;
    if (BRICKS && !voxel_occupied(p, vx, vy, vz)) {
      res[j] = -limit;
      continue;
    }
    if (vx >= p.X || vy >= p.Y || vz >= p.Z) {  // padding voxel of a partial tile
      res[j] = -limit;
      continue;
    }
    const float px = ((float)vx + 0.5f) * p.stepX;
    float tsd = limit, wsum = 0.0f;
    for (int i = 0; i < p.N; ++i) {
      const float3 pc = tex3d_xyz(p.lut[i], p.rx[i], p.ry[i], p.rz[i], p.zoff[i], px, py, pz);
      float ax, ay;
      const int jx = footprint(pc.x, p.W, ax), jy = footprint(pc.y, p.H, ay);
      uint2 p00, p10, p01, p11;
      fetch_global(p.frame[i], p.W, p.H, jx, jy, p00, p10, p01, p11);
      fold_taps(p00, p10, p01, p11, ax, ay, pc.z, limit, tsd, wsum);
    }
    res[j] = tsd;
  }
  *out = make_float4(res[0], res[1], res[2], res[3]);
}

// The same sweep with the LUT block of a tile staged in LDS (RGBDR_FLAG_NO_RESAMPLE at a LUT that is not much finer than
// the grid).  The plain kernel above gathers 8 x 16 B per voxel and sensor from global memory in tile order: 15.8 GB of HBM
// traffic for a 512^3 sweep whose four 256^3 LUTs hold 1.07 GB (profiles/r06_pmc_generic.json), 9 x the time of the
// resampled sweep.  Here the texels a tile's 512 voxels can touch -- the index box from the first voxel's lower to the last
// voxel's upper texel, per sensor -- are loaded once, row by row (.xyz only: 12 B), and the eight taps come from LDS with
// tex3d_xyz's expressions (bit-identical).  The fold over the sensors stays sequential per voxel.  bx * by * bz: the largest
// box over all tiles and sensors, found on the host (launch_integrate).
template <bool BRICKS>
__global__ __launch_bounds__(128) void k_integrate_generic_lds(IntegrateParams p, int box_texels)
{
  extern __shared__ float s_box[];  // [nz][ny][nx][3] of the current sensor
  unsigned tile = blockIdx.x;
  if (p.order_chunk) {  // XCD-aware order like the tiled sweep: neighbouring tiles share LUT rows in one XCD's L2
    const unsigned xcd = tile & 7u, k = tile >> 3, chunk = p.order_chunk;
    tile = (k / chunk) * 8u * chunk + xcd * chunk + k % chunk;
  }
  const int q = threadIdx.x;
  const int lz = q >> 4, ly = (q >> 1) & 7, lx0 = (q & 1) * 4;
  const int tx = tile % p.TX;
  const int ty = (tile / p.TX) % p.TY;
  const int tzl = tile / (p.TX * p.TY);
  const int vz = (p.tz0 + tzl) * kTile + lz, vy = ty * kTile + ly, vx0 = tx * kTile + lx0;
  float4* out = reinterpret_cast<float4*>(p.tsdf + (size_t)tile * kTileVoxels) + q;
  const float limit = p.limit;
  const float pz = ((float)vz + 0.5f) * p.stepZ, py = ((float)vy + 0.5f) * p.stepY;
  float tsd[4], wsum[4];
  bool live[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int vx = vx0 + j;
    tsd[j] = limit;
    wsum[j] = 0.0f;
    live[j] = !(BRICKS && !voxel_occupied(p, vx, vy, vz)) && !(vx >= p.X || vy >= p.Y || vz >= p.Z);
  }
  // the tile's voxel range along every axis (a partial tile ends at the grid's last voxel)
  const int tvx0 = tx * kTile, tvy0 = ty * kTile, tvz0 = (p.tz0 + tzl) * kTile;
  const int tvx1 = min(tvx0 + kTile - 1, p.X - 1), tvy1 = min(tvy0 + kTile - 1, p.Y - 1), tvz1 = min(tvz0 + kTile - 1, p.Z - 1);
  for (int i = 0; i < p.N; ++i) {
    const int rx = p.rx[i], ry = p.ry[i], rz = p.rz[i];
    // index box: monotone in the voxel index, so the first voxel's i0 and the last voxel's i1 bound every tap
    const int X0 = axis_linear(((float)tvx0 + 0.5f) * p.stepX, rx).i0, X1 = axis_linear(((float)tvx1 + 0.5f) * p.stepX, rx).i1;
    const int Y0 = axis_linear(((float)tvy0 + 0.5f) * p.stepY, ry).i0, Y1 = axis_linear(((float)tvy1 + 0.5f) * p.stepY, ry).i1;
    const int Z0 = axis_linear(((float)tvz0 + 0.5f) * p.stepZ, rz).i0, Z1 = axis_linear(((float)tvz1 + 0.5f) * p.stepZ, rz).i1;
    const int nx = X1 - X0 + 1, ny = Y1 - Y0 + 1, nz = Z1 - Z0 + 1;
    const int n = nx * ny * nz;
    __syncthreads();  // the previous sensor's taps are all read
    if (n <= box_texels) {
      const float4* __restrict__ lut = p.lut[i];
      for (int t = q; t < n; t += 128) {
        const int x = t % nx, y = (t / nx) % ny, z = t / (nx * ny);
        const float4 v = lut[((size_t)(Z0 + z - p.zoff[i]) * ry + (Y0 + y)) * rx + (X0 + x)];
        s_box[3 * t] = v.x;
        s_box[3 * t + 1] = v.y;
        s_box[3 * t + 2] = v.z;
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (!live[j]) continue;
      const float px = ((float)(vx0 + j) + 0.5f) * p.stepX;
      float3 pc;
      if (n <= box_texels) {
        const Axis AX = axis_linear(px, rx), AY = axis_linear(py, ry), AZ = axis_linear(pz, rz);
        auto at = [&](int x, int y, int z) {
          const float* t = s_box + 3 * (((z - Z0) * ny + (y - Y0)) * nx + (x - X0));
          return make_float3(t[0], t[1], t[2]);
        };
        const float3 c00 = lerp3(at(AX.i0, AY.i0, AZ.i0), at(AX.i1, AY.i0, AZ.i0), AX.a), c10 = lerp3(at(AX.i0, AY.i1, AZ.i0), at(AX.i1, AY.i1, AZ.i0), AX.a);
        const float3 c01 = lerp3(at(AX.i0, AY.i0, AZ.i1), at(AX.i1, AY.i0, AZ.i1), AX.a), c11 = lerp3(at(AX.i0, AY.i1, AZ.i1), at(AX.i1, AY.i1, AZ.i1), AX.a);
        pc = lerp3(lerp3(c00, c10, AY.a), lerp3(c01, c11, AY.a), AZ.a);
      } else {  // (a tile whose box outgrows LDS: a LUT much finer than the grid)
        pc = tex3d_xyz(p.lut[i], rx, ry, rz, p.zoff[i], px, py, pz);
      }
      float ax, ay;
      const int jx = footprint(pc.x, p.W, ax), jy = footprint(pc.y, p.H, ay);
      uint2 p00, p10, p01, p11;
      fetch_global(p.frame[i], p.W, p.H, jx, jy, p00, p10, p01, p11);
      fold_taps(p00, p10, p01, p11, ax, ay, pc.z, limit, tsd[j], wsum[j]);
    }
  }
  float res[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) res[j] = live[j] ? tsd[j] : -limit;
  *out = make_float4(res[0], res[1], res[2], res[3]);
}

// largest LUT index box of a tile along one axis (the kernel's own arithmetic, on the host: same floats)
static int generic_box_extent(int tiles, int voxels, float step, int r)
{
  auto i01 = [&](int v, bool upper) {
    const float t = ((float)v + 0.5f) * step * (float)r - 0.5f;
    const float f = std::floor(t);
    const int j = (int)std::fmin(std::fmax(f, -1.0f), (float)r);
    const int k = upper ? j + 1 : j;
    return k < 0 ? 0 : (k > r - 1 ? r - 1 : k);
  };
  int ext = 1;
  for (int t = 0; t < tiles; ++t) {
    const int v0 = t * kTile, v1 = std::min(v0 + kTile - 1, voxels - 1);
    ext = std::max(ext, i01(v1, true) - i01(v0, false) + 1);
  }
  return ext;
}

// The full sweep as p.launches launches over consecutive block ranges (whole rounds of 8 chunks each).  Between two of
// them the queue drains: the moment a kernel waiting on ANOTHER queue gets its workgroups placed -- next to one launch
// that refills every wave slot as it frees, RCCL's gather kernel sits until the sweep ends (profiles/r05_notes).
template <typename Kernel>
static void launch_in_parts(Kernel kernel, IntegrateParams p, unsigned ntiles, hipStream_t s)
{
  const unsigned unit = 8u * (p.order_chunk ? p.order_chunk : 1u), units = ntiles / unit;
  const unsigned parts = p.launches > 1 ? (p.launches < units ? p.launches : (units ? units : 1u)) : 1u;
  for (unsigned i = 0; i < parts; ++i) {
    const unsigned b0 = (unsigned)((uint64_t)units * i / parts) * unit;
    const unsigned b1 = i + 1 == parts ? ntiles : (unsigned)((uint64_t)units * (i + 1) / parts) * unit;
    if (b1 == b0) continue;
    p.block_base = b0;
    hipLaunchKernelGGL(kernel, dim3(b1 - b0), dim3(128), 0, s, p);
  }
}

template <int N>
static void launch_tiled_n(const IntegrateParams& p, unsigned ntiles, hipStream_t s)
{
  if (p.use_bricks) {
    launch_brick_sweep(p, ntiles, s);  // kernels_bricks.hip
    return;
  }
#ifdef RGBDR_DEV_KNOBS  // developer A/B builds only (make EXTRA=-DRGBDR_DEV_KNOBS): the shipped launch path reads no environment
  // RGBDR_INTEGRATE_GROUP=2 folds 3 or 4 sensors in two groups
  static const int maxg = getenv("RGBDR_INTEGRATE_GROUP") ? atoi(getenv("RGBDR_INTEGRATE_GROUP")) : 4;
  if (maxg == 2 && (N == 3 || N == 4)) {
    hipLaunchKernelGGL((k_integrate_tiled<N, 2>), dim3(ntiles), dim3(128), 0, s, p);
    return;
  }
  // RGBDR_NT=0 uses temporal loads/stores for the LUT / TSDF streams
  static const bool temporal = getenv("RGBDR_NT") && !atoi(getenv("RGBDR_NT"));
  if (temporal && N == 4) {
    hipLaunchKernelGGL((k_integrate_tiled<N, 4, false>), dim3(ntiles), dim3(128), 0, s, p);
    return;
  }
#endif
  if (p.elide_stores)
    launch_in_parts(k_integrate_tiled<N, 4, true, true>, p, ntiles, s);
  else if (p.stage_lo || p.stage_hi)
    launch_in_parts(k_integrate_tiled<N, 4, true, false, true>, p, ntiles, s);
  else
    launch_in_parts(k_integrate_tiled<N>, p, ntiles, s);
}

// true when launch_integrate's kernel itself fills the halo staging buffers (plain full sweep of a
// 1:1 / resampled LUT); for every other sweep the caller copies the layers afterwards
bool integrate_stages_halo(const IntegrateParams& p, bool one_to_one)
{
  if (!one_to_one || p.use_bricks || p.elide_stores || p.skip_background) return false;
#ifdef RGBDR_DEV_KNOBS
  static const bool knobs = getenv("RGBDR_INTEGRATE_GROUP") || getenv("RGBDR_NT");
  return !knobs;
#else
  return true;
#endif
}

void launch_integrate(const IntegrateParams& p_in, bool one_to_one, hipStream_t s)
{
  IntegrateParams p = p_in;
  const unsigned ntiles = (unsigned)p.TX * p.TY * p.ntz;
  // full sweep: chunk = one x-row of tiles when the grid divides evenly (measured
  // 3-8 % faster than identity / one contiguous run per XCD); brick-skipping sweep:
  // identity (most blocks only clear their tile; chunked order measured 40 % slower).
  unsigned chunk = p.use_bricks ? 0u : (unsigned)p.TX;
#ifdef RGBDR_DEV_KNOBS  // RGBDR_TILE_CHUNK overrides (0 = identity, -1 = ntiles/8)
  static const char* chunk_env = getenv("RGBDR_TILE_CHUNK");
  if (chunk_env) chunk = atoi(chunk_env) < 0 ? ntiles / 8 : (unsigned)atoi(chunk_env);
#endif
  p.order_chunk = (chunk && ntiles % (8 * chunk) == 0) ? chunk : 0;
  if (!one_to_one) {
    // the LUT box of a tile in LDS when it fits (k_integrate_generic_lds); RGBDR_GENERIC_GLOBAL=1: the plain gathers
    static const bool global_only = getenv("RGBDR_GENERIC_GLOBAL") != nullptr;
    int box = 0;
    for (int i = 0; i < p.N && !global_only; ++i) {
      const int e = generic_box_extent(p.TX, p.X, p.stepX, p.rx[i]) * generic_box_extent(p.TY, p.Y, p.stepY, p.ry[i]) *
                    generic_box_extent(p.tz0 + p.ntz, p.Z, p.stepZ, p.rz[i]);
      box = std::max(box, e);
    }
    const size_t lds = (size_t)box * 3 * sizeof(float);
    if (box > 0 && lds <= 96 * 1024) {
      if (p.use_bricks) {
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k_integrate_generic_lds<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((k_integrate_generic_lds<true>), dim3(ntiles), dim3(128), lds, s, p, box);
      } else {
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k_integrate_generic_lds<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((k_integrate_generic_lds<false>), dim3(ntiles), dim3(128), lds, s, p, box);
      }
      return;
    }
    if (p.use_bricks)
      hipLaunchKernelGGL((k_integrate_generic<true>), dim3(ntiles), dim3(128), 0, s, p);
    else
      hipLaunchKernelGGL((k_integrate_generic<false>), dim3(ntiles), dim3(128), 0, s, p);
    return;
  }
  switch (p.N) {
    case 1: launch_tiled_n<1>(p, ntiles, s); break;
    case 2: launch_tiled_n<2>(p, ntiles, s); break;
    case 3: launch_tiled_n<3>(p, ntiles, s); break;
    case 4: launch_tiled_n<4>(p, ntiles, s); break;
    case 5: launch_tiled_n<5>(p, ntiles, s); break;
    case 6: launch_tiled_n<6>(p, ntiles, s); break;
    case 7: launch_tiled_n<7>(p, ntiles, s); break;
    default: launch_tiled_n<8>(p, ntiles, s); break;
  }
}

// ---------------------------------------------------------------------------
// tile-linear -> x-fastest linear (readback helper; one thread per voxel)
__global__ void k_detile(const float* __restrict__ tiled, float* __restrict__ linear, int X, int Y, int TX, int TY,
                         int tz0, int vz0, int vz1)
{
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y, z = vz0 + blockIdx.z;
  if (x >= X || z >= vz1) return;
  const size_t tile = ((size_t)(z / kTile - tz0) * TY + (y / kTile)) * TX + (x / kTile);
  const int in = (z & 7) * 64 + (y & 7) * 8 + (x & 7);
  linear[((size_t)(z - vz0) * Y + y) * X + x] = tiled[tile * kTileVoxels + in];
}
void launch_detile(const float* tiled, float* linear, int X, int Y, int TX, int TY, int tz0, int vz0, int vz1,
                   hipStream_t s)
{
  dim3 grid((X + 127) / 128, Y, vz1 - vz0);
  hipLaunchKernelGGL(k_detile, grid, dim3(128), 0, s, tiled, linear, X, Y, TX, TY, tz0, vz0, vz1);
}

// x-fastest RGBA volume (z rows [src_z0, ...) resident at src) -> tiled planes
__global__ __launch_bounds__(128) void k_tile_lut(const float4* __restrict__ src, int X, int Y, int Z, int src_z0,
                                                  int TX, int TY, int tz0, int sensor, int N,
                                                  float* __restrict__ dst)
{
  const unsigned tile = blockIdx.x;
  const int q = threadIdx.x;
  const int lz = q >> 4, ly = (q >> 1) & 7, lx0 = (q & 1) * 4;
  const int tx = tile % TX, ty = (tile / TX) % TY, tzl = tile / (TX * TY);
  const int vz = (tz0 + tzl) * kTile + lz, vy = ty * kTile + ly, vx0 = tx * kTile + lx0;
  float u[4], v[4], d[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int vx = vx0 + j;
    float4 t = make_float4(-1.0f, -1.0f, -1.0f, -1.0f);
    if (vx < X && vy < Y && vz < Z) t = src[((size_t)(vz - src_z0) * Y + vy) * X + vx];
    u[j] = t.x;
    v[j] = t.y;
    d[j] = t.z;
  }
  float4* o = reinterpret_cast<float4*>(dst + ((size_t)tile * N + sensor) * 3 * kTileVoxels) + q;
  o[0] = make_float4(u[0], u[1], u[2], u[3]);
  o[kTileVoxels / 4] = make_float4(v[0], v[1], v[2], v[3]);
  o[2 * (kTileVoxels / 4)] = make_float4(d[0], d[1], d[2], d[3]);
}
// ---------------------------------------------------------------------------
// Arena placement probe.  The sweep time of the integrate kernel is a stable property of
// where the driver placed the LUT arena (the same binary measures 1.06 / 1.14 / 1.19 ms on
// different hipMalloc results; the TSDF buffer and the tile order do not matter --
// profiles/probes_src/pair_probe.hip).  This kernel replays the LUT stream
// of k_integrate_tiled (one 128-thread block per tile, per_tile_v4 contiguous 16-byte loads,
// XCD-chunked order) so that rgbdr can time candidate arenas at upload and keep the fastest.
typedef float probe_v4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(128) void k_arena_probe(const probe_v4* __restrict__ arena, unsigned ntiles, unsigned per_tile_v4,
                                                     unsigned chunk, float* __restrict__ sink)
{
  unsigned b = blockIdx.x;
  if (chunk) {
    const unsigned xcd = b & 7u, slot = b >> 3, span = chunk * 8u;
    b = (slot / chunk) * span + xcd * chunk + slot % chunk;
  }
  const probe_v4* q = arena + (size_t)b * per_tile_v4;
  probe_v4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
  for (unsigned k = threadIdx.x; k < per_tile_v4; k += 128) acc += __builtin_nontemporal_load(q + k);
  // the TSDF tile store of the real kernel (the effect only shows with the write stream present)
  __builtin_nontemporal_store(acc, (probe_v4*)sink + (size_t)b * (kTileVoxels / 4) + threadIdx.x);
}

float probe_arena_ms(const float* arena, size_t ntiles, int N, int TX, float* sink, hipStream_t s)
{
  const unsigned per_tile = (unsigned)N * 3u * (kTileVoxels / 4);
  const unsigned chunk = (TX > 0 && ntiles % (8u * (unsigned)TX) == 0) ? (unsigned)TX : 0u;
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess) return -1.0f;
  if (hipEventCreate(&e1) != hipSuccess) {
    (void)hipEventDestroy(e0);
    return -1.0f;
  }
  float ms = -1.0f;
  hipLaunchKernelGGL(k_arena_probe, dim3((unsigned)ntiles), dim3(128), 0, s, (const probe_v4*)arena, (unsigned)ntiles, per_tile,
                     chunk, sink);
  (void)hipEventRecord(e0, s);
  const int reps = 3;
  for (int r = 0; r < reps; ++r)
    hipLaunchKernelGGL(k_arena_probe, dim3((unsigned)ntiles), dim3(128), 0, s, (const probe_v4*)arena, (unsigned)ntiles,
                       per_tile, chunk, sink);
  (void)hipEventRecord(e1, s);
  if (hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess) ms /= reps;
  else ms = -1.0f;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return ms;
}

void launch_tile_lut(const float4* src, int X, int Y, int Z, int src_z0, int TX, int TY, int tz0, int ntz, int sensor,
                     int N, float* dst, hipStream_t s)
{
  hipLaunchKernelGGL(k_tile_lut, dim3((unsigned)TX * TY * ntz), dim3(128), 0, s, src, X, Y, Z, src_z0, TX, TY, tz0,
                     sensor, N, dst);
}

// Inverse LUT of any resolution -> grid layout: evaluates the LINEAR lookup
// texture(cv_xyz_inv[i], position) of tsdf_integration.vs:31 once per voxel centre
// at upload time (LUT and grid are static between frames, so the per-frame result
// is bit-identical) and stores it in the tiled planes the 1:1 kernel streams.
__global__ __launch_bounds__(128) void k_resample_lut(const float4* __restrict__ src, int rx, int ry, int rz, int zoff,
                                                      int X, int Y, int Z, int TX, int TY, int tz0, int sensor, int N,
                                                      float* __restrict__ dst)
{
  const unsigned tile = blockIdx.x;
  const int q = threadIdx.x;
  const int lz = q >> 4, ly = (q >> 1) & 7, lx0 = (q & 1) * 4;
  const int tx = tile % TX, ty = (tile / TX) % TY, tzl = tile / (TX * TY);
  const int vz = (tz0 + tzl) * kTile + lz, vy = ty * kTile + ly, vx0 = tx * kTile + lx0;
  const float stepX = 1.0f / (float)X, stepY = 1.0f / (float)Y, stepZ = 1.0f / (float)Z;
  const float pz = ((float)vz + 0.5f) * stepZ, py = ((float)vy + 0.5f) * stepY;
  float u[4], v[4], d[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int vx = vx0 + j;
    u[j] = v[j] = d[j] = -1.0f;
    if (vx >= X || vy >= Y || vz >= Z) continue;  // padding voxel of a partial tile
    const float px = ((float)vx + 0.5f) * stepX;
    const float3 pc = tex3d_xyz(src, rx, ry, rz, zoff, px, py, pz);
    u[j] = pc.x;
    v[j] = pc.y;
    d[j] = pc.z;
  }
  float4* o = reinterpret_cast<float4*>(dst + ((size_t)tile * N + sensor) * 3 * kTileVoxels) + q;
  o[0] = make_float4(u[0], u[1], u[2], u[3]);
  o[kTileVoxels / 4] = make_float4(v[0], v[1], v[2], v[3]);
  o[2 * (kTileVoxels / 4)] = make_float4(d[0], d[1], d[2], d[3]);
}
void launch_resample_lut(const float4* src, int rx, int ry, int rz, int zoff, int X, int Y, int Z, int TX, int TY,
                         int tz0, int ntz, int sensor, int N, float* dst, hipStream_t s)
{
  hipLaunchKernelGGL(k_resample_lut, dim3((unsigned)TX * TY * ntz), dim3(128), 0, s, src, rx, ry, rz, zoff, X, Y, Z, TX,
                     TY, tz0, sensor, N, dst);
}

__global__ void k_untile_lut(const float* __restrict__ tiled, int X, int Y, int TX, int TY, int tz0, int vz0, int vz1,
                             int sensor, int N, float4* __restrict__ dst)
{
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y, z = vz0 + blockIdx.z;
  if (x >= X || z >= vz1) return;
  const size_t tile = ((size_t)(z / kTile - tz0) * TY + (y / kTile)) * TX + (x / kTile);
  const int in = (z & 7) * 64 + (y & 7) * 8 + (x & 7);
  const float* b = tiled + (tile * N + sensor) * 3 * kTileVoxels;
  dst[((size_t)(z - vz0) * Y + y) * X + x] = make_float4(b[in], b[kTileVoxels + in], b[2 * kTileVoxels + in], 0.0f);
}
void launch_untile_lut(const float* tiled, int X, int Y, int TX, int TY, int tz0, int vz0, int vz1, int sensor, int N,
                       float4* dst, hipStream_t s)
{
  dim3 grid((X + 127) / 128, Y, vz1 - vz0);
  hipLaunchKernelGGL(k_untile_lut, grid, dim3(128), 0, s, tiled, X, Y, TX, TY, tz0, vz0, vz1, sensor, N, dst);
}

// ---------------------------------------------------------------------------
// Benchmark support: analytic inverse LUT of a pinhole sensor written straight
// into the tiled planes (SURVEY.md section 8d "LUT generation ... on-device").
__global__ __launch_bounds__(128) void k_synth_inverse(rgbdr_pinhole cam, int W, int H, float3 bmin, float3 bext, int X,
                                                       int Y, int Z, int TX, int TY, int tz0, int sensor, int N,
                                                       float* __restrict__ dst)
{
  const unsigned tile = blockIdx.x;
  const int q = threadIdx.x;
  const int lz = q >> 4, ly = (q >> 1) & 7, lx0 = (q & 1) * 4;
  const int tx = tile % TX, ty = (tile / TX) % TY, tzl = tile / (TX * TY);
  const int vz = (tz0 + tzl) * kTile + lz, vy = ty * kTile + ly, vx0 = tx * kTile + lx0;
  float u[4], v[4], d[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int vx = vx0 + j;
    u[j] = v[j] = d[j] = -1.0f;
    if (vx >= X || vy >= Y || vz >= Z) continue;
    const float wx = bmin.x + (((float)vx + 0.5f) / (float)X) * bext.x;
    const float wy = bmin.y + (((float)vy + 0.5f) / (float)Y) * bext.y;
    const float wz = bmin.z + (((float)vz + 0.5f) / (float)Z) * bext.z;
    const float rx = wx - cam.cam_pos[0], ry = wy - cam.cam_pos[1], rz = wz - cam.cam_pos[2];
    const float xc = rx * cam.right[0] + ry * cam.right[1] + rz * cam.right[2];
    const float yc = rx * cam.up[0] + ry * cam.up[1] + rz * cam.up[2];
    const float zc = rx * cam.forward[0] + ry * cam.forward[1] + rz * cam.forward[2];
    if (!(zc >= cam.depth_min && zc <= cam.depth_max)) continue;
    const float pxl = cam.fx * xc / zc + cam.cx, pyl = cam.fy * yc / zc + cam.cy;
    if (!(pxl >= 0.0f && pxl < (float)W && pyl >= 0.0f && pyl < (float)H)) continue;
    u[j] = pxl / (float)W;
    v[j] = pyl / (float)H;
    d[j] = (zc - cam.depth_min) / (cam.depth_max - cam.depth_min);
  }
  float4* o = reinterpret_cast<float4*>(dst + ((size_t)tile * N + sensor) * 3 * kTileVoxels) + q;
  o[0] = make_float4(u[0], u[1], u[2], u[3]);
  o[kTileVoxels / 4] = make_float4(v[0], v[1], v[2], v[3]);
  o[2 * (kTileVoxels / 4)] = make_float4(d[0], d[1], d[2], d[3]);
}
void launch_synth_inverse(const rgbdr_pinhole& cam, int W, int H, const float bbox_min[3], const float bbox_max[3],
                          int X, int Y, int Z, int TX, int TY, int tz0, int ntz, int sensor, int N, float* dst,
                          hipStream_t s)
{
  const float3 bmin = make_float3(bbox_min[0], bbox_min[1], bbox_min[2]);
  const float3 bext = make_float3(bbox_max[0] - bbox_min[0], bbox_max[1] - bbox_min[1], bbox_max[2] - bbox_min[2]);
  hipLaunchKernelGGL(k_synth_inverse, dim3((unsigned)TX * TY * ntz), dim3(128), 0, s, cam, W, H, bmin, bext, X, Y, Z,
                     TX, TY, tz0, sensor, N, dst);
}

}  // namespace rgbdr
