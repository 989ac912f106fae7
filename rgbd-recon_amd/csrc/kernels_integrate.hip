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
                                                float* tsd, float* wsum, unsigned actions = 0u)
{
  int wx0[CNT], wy0[CNT];
  unsigned act[CNT];
#pragma unroll
  for (int i = 0; i < CNT; ++i) {
    const int d = p.win[(size_t)tile * ntot + s0 + i];
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

// 1:1 LUT.  128 threads (2 wavefronts) sweep one 8x8x8 tile; thread q owns voxels
// x0..x0+3 of row (y,z).  Every global load of a sensor group -- the 3 LUT planes
// per sensor (16 B per lane, fully coalesced) and the 16x16 frame window per sensor
// the tile projects into (origins precomputed per tile at LUT upload) -- is issued
// before that group's barrier, so one memory latency is paid per group; the 2x2
// footprints of all 512 voxels are then served from LDS.  More than 4 sensors are
// folded in two groups (the running tsd / weight stay in registers), which keeps
// the kernel at <= ~100 VGPRs for every N.
// One tile: BRICKS marks the voxels of unoccupied bricks -limit after the fold.
template <int N, int MAXG, bool NT, bool ELIDE = false, bool STAGE = false>
__device__ __forceinline__ void integrate_tile(const IntegrateParams& p, unsigned tile, uint2 (*win)[kWin * kWinPitch])
{
  constexpr int G1 = N <= MAXG ? N : (N + 1) / 2;  // first group
  constexpr int G2 = N - G1;                       // second group (0 for N <= MAXG)
  const int q = threadIdx.x;
  float4* out = reinterpret_cast<float4*>(p.tsdf + (size_t)tile * kTileVoxels) + q;
  const float limit = p.limit;
  float tsd[4] = {limit, limit, limit, limit};
  float wsum[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  integrate_group<G1, NT, false>(p, tile, q, 0, N, win, false, limit, tsd, wsum);
  if (G2 > 0) integrate_group<(G2 > 0 ? G2 : 1), NT, false>(p, tile, q, G1, N, win, true, limit, tsd, wsum);
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
  if (STAGE) {  // boundary tile layers of a Z slab go to the halo staging buffers as well
    const unsigned per_layer = (unsigned)(p.TX * p.TY), layer = tile / per_layer;
    const float4 r4 = make_float4(tsd[0], tsd[1], tsd[2], tsd[3]);
    if (p.stage_lo && layer < (unsigned)p.stage_layers) reinterpret_cast<float4*>(p.stage_lo + (size_t)tile * kTileVoxels)[q] = r4;
    const unsigned first_hi = (unsigned)(p.ntz - p.stage_layers);
    if (p.stage_hi && layer >= first_hi)
      reinterpret_cast<float4*>(p.stage_hi + (size_t)(tile - first_hi * per_layer) * kTileVoxels)[q] = r4;
  }
}

// Full sweep: one block per tile.
template <int N, int MAXG = 4, bool NT = true, bool ELIDE = false, bool STAGE = false>
__global__ __launch_bounds__(128) void k_integrate_tiled(IntegrateParams p)
{
  constexpr int G1 = N <= MAXG ? N : (N + 1) / 2;
  __shared__ uint2 win[G1][kWin * kWinPitch];
  // XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch).
  // Each XCD takes chunks of `order_chunk` consecutive tiles (neighbouring tiles
  // project into overlapping frame windows -> hits in that XCD's L2), and the 8
  // XCDs work on 8 adjacent chunks at a time so the concurrent LUT streams stay
  // close together in memory.
  unsigned tile = blockIdx.x;
  if (p.order_chunk) {
    const unsigned xcd = blockIdx.x & 7u, idx = blockIdx.x >> 3;
    const unsigned chunk = idx / p.order_chunk, within = idx - chunk * p.order_chunk;
    tile = (chunk * 8u + xcd) * p.order_chunk + within;
  }
  integrate_tile<N, MAXG, NT, ELIDE, STAGE>(p, tile, win);
}

// ---------------------------------------------------------------------------
// Brick-skipping sweep, second half: persistent blocks walk the list of tiles that touch an occupied brick
// (k_brick_clear wrote -limit everywhere else and built the list).
//
// The list is short (a few tiles per block) and a tile's life is a chain of dependent loads, so the sweep is
// bound by latency, not by HBM or the VALU (per tile, measured with s_memtime: 2.9 us waiting for loads and
// 6.3 us in the fold with four wavefronts per SIMD; the fold alone takes 5 us with two).  What is done about it:
//   * everything a tile's loads depend on -- list entry, window origins, brick ranges: block-uniform words --
//     is fetched one tile ahead, through the constant address space (scalar loads that nothing waits for);
//   * a tile whose bricks are all occupied (bit 31 of the list entry, k_brick_clear) skips the occupancy test;
//   * for the others the range words are loaded with the LUT planes and the mask bytes of the few bricks the
//     tile touches go to LDS with the frame windows, so occupancy adds no load latency of its own.
// Five wavefronts per SIMD (96 VGPRs), 10 blocks per CU.
// Tried and dropped (profiles/r02_notes): issuing the next tile's loads before folding the current one (two
// wavefronts per SIMD instead of four: 0.082 vs 0.072 ms), folding two or four voxels of a lane together for
// instruction-level parallelism (spills at four wavefronts per SIMD: 0.081 / 0.146 ms), 3 / 5 / 6 wavefronts
// per SIMD (0.078 / 0.096 / 0.128 ms).
constexpr int kOccSide = 4;  // brick mask cache of a partial tile: up to 4^3 bricks (others: global path)

template <int GS>
struct StageLoads {  // per-thread loads of one stage (= one sensor group of one tile): LUT planes + frame window texels
  float4 U[GS], V[GS], D[GS];
  uint2 ta[GS], tb[GS];
};
struct OccLoads {    // per-thread occupancy inputs of a partial tile
  uint32_t ex[4], ey, ez;
  uint8_t mask;
};
template <int N>
struct TileWords {   // block-uniform words of one list entry
  unsigned entry;    // tile | whole << 31; 0xffffffff: no such stage
  int org[N];        // window origins (int16 x | int16 y << 16) per sensor
  uint32_t rx, ry, rz;  // BrickTables::tile ranges (partial tiles)
};

// `entry` was fetched an iteration earlier (0xffffffff: past the end of the list), so nothing here waits on a load
template <int N>
__device__ __forceinline__ void load_tile_words(const IntegrateParams& p, unsigned entry, TileWords<N>& w)
{
  w.entry = entry;
  w.rx = w.ry = w.rz = 0x0000ffffu;
#pragma unroll
  for (int s = 0; s < N; ++s) w.org[s] = 0;
  if (entry == 0xffffffffu) return;
  const unsigned tile = entry & 0x7fffffffu;
#pragma unroll
  for (int s = 0; s < N; ++s) w.org[s] = ro(p.win)[(size_t)tile * N + s];
  if (!(entry >> 31)) {
    w.rx = ro(p.tbx)[tile % p.TX];
    w.ry = ro(p.tby)[(tile / p.TX) % p.TY];
    w.rz = ro(p.tbz)[p.tz0 + tile / (p.TX * p.TY)];
  }
}

// does the LDS mask cache serve this (partial) tile?
template <int N>
__device__ __forceinline__ bool occ_cached(const IntegrateParams& p, const TileWords<N>& w, uint32_t* lo, uint32_t* cnt)
{
  lo[0] = w.rx & 0xffffu, lo[1] = w.ry & 0xffffu, lo[2] = w.rz & 0xffffu;
  const uint32_t hx = w.rx >> 16, hy = w.ry >> 16, hz = w.rz >> 16;
  cnt[0] = hx - lo[0] + 1u, cnt[1] = hy - lo[1] + 1u, cnt[2] = hz - lo[2] + 1u;
  return !(p.ovx | p.ovy) && lo[0] <= hx && lo[1] <= hy && lo[2] <= hz && cnt[0] <= (uint32_t)kOccSide &&
         cnt[1] <= (uint32_t)kOccSide && cnt[2] <= (uint32_t)kOccSide;
}

template <int N, int GS>
__device__ __forceinline__ void issue_stage(const IntegrateParams& p, const TileWords<N>& w, int g, int q, StageLoads<GS>& L)
{
  typedef float v4f __attribute__((ext_vector_type(4)));
  const unsigned tile = w.entry & 0x7fffffffu;
  const int s0 = g * GS;
  const v4f* l = reinterpret_cast<const v4f*>(p.lut_tiled + ((size_t)tile * N + s0) * (3 * kTileVoxels)) + q;
  const int wr = q >> 3, wc = (q & 7) * 2;
#pragma unroll
  for (int i = 0; i < GS; ++i) {
    if (s0 + i < N) {
      const v4f u = __builtin_nontemporal_load(&l[(i * 3 + 0) * (kTileVoxels / 4)]);
      const v4f v = __builtin_nontemporal_load(&l[(i * 3 + 1) * (kTileVoxels / 4)]);
      const v4f d = __builtin_nontemporal_load(&l[(i * 3 + 2) * (kTileVoxels / 4)]);
      L.U[i] = make_float4(u.x, u.y, u.z, u.w);
      L.V[i] = make_float4(v.x, v.y, v.z, v.w);
      L.D[i] = make_float4(d.x, d.y, d.z, d.w);
    }
  }
#pragma unroll
  for (int i = 0; i < GS; ++i) {
    if (s0 + i < N) {
      const int wx0 = (int)(short)(w.org[s0 + i] & 0xffff), wy0 = (int)(short)(w.org[s0 + i] >> 16);
      const int row = clampi(wy0 + wr, 0, p.H - 1) * p.W;
      L.ta[i] = p.frame[s0 + i][row + clampi(wx0 + wc, 0, p.W - 1)];
      L.tb[i] = p.frame[s0 + i][row + clampi(wx0 + wc + 1, 0, p.W - 1)];
    }
  }
}

// occupancy inputs of a partial tile, issued with its first stage
template <int N>
__device__ __forceinline__ void issue_occ(const IntegrateParams& p, const TileWords<N>& w, int q, OccLoads& o)
{
  const unsigned tile = w.entry & 0x7fffffffu;
  const int lz = q >> 4, ly = (q >> 1) & 7, lx0 = (q & 1) * 4;
  const int vx = (int)(tile % p.TX) * kTile + lx0, vy = (int)((tile / p.TX) % p.TY) * kTile + ly;
  const int vz = (p.tz0 + (int)(tile / (p.TX * p.TY))) * kTile + lz;
#pragma unroll
  for (int j = 0; j < 4; ++j) o.ex[j] = vx + j < p.X ? p.vbx[vx + j] : 0x0000ffffu;
  o.ey = vy < p.Y ? p.vby[vy] : 0x0000ffffu;
  o.ez = vz < p.Z ? p.vbz[vz] : 0x0000ffffu;
  uint32_t lo[3], cnt[3];
  o.mask = 0;
  if (occ_cached(p, w, lo, cnt) && (uint32_t)q < cnt[0] * cnt[1] * cnt[2]) {
    const uint32_t bx = (uint32_t)q % cnt[0], by = ((uint32_t)q / cnt[0]) % cnt[1], bz = (uint32_t)q / (cnt[0] * cnt[1]);
    o.mask = p.brick_mask[((size_t)(lo[2] + bz) * p.by + (lo[1] + by)) * p.bx + (lo[0] + bx)];
  }
}

template <int N, int GS>
__device__ __forceinline__ void fold_stage(const IntegrateParams& p, const TileWords<N>& w, int g, int q,
                                           const StageLoads<GS>& L, uint2 (*win)[kWin * kWinPitch], float limit, float* tsd,
                                           float* wsum)
{
  const int s0 = g * GS;
  const int wr = q >> 3, wc = (q & 7) * 2;
  __syncthreads();  // the previous stage's footprints (and mask cache) are all read
#pragma unroll
  for (int i = 0; i < GS; ++i) {
    if (s0 + i < N) {
      win[i][wr * kWinPitch + wc] = L.ta[i];
      win[i][wr * kWinPitch + wc + 1] = L.tb[i];
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < GS; ++i) {
    if (s0 + i < N) {
      const uint2* frame = p.frame[s0 + i];
      const int wx0 = (int)(short)(w.org[s0 + i] & 0xffff), wy0 = (int)(short)(w.org[s0 + i] >> 16);
        fold_voxel_window(win[i], wx0, wy0, frame, p.W, p.H, L.U[i].x, L.V[i].x, L.D[i].x, limit, tsd[0], wsum[0]);
        fold_voxel_window(win[i], wx0, wy0, frame, p.W, p.H, L.U[i].y, L.V[i].y, L.D[i].y, limit, tsd[1], wsum[1]);
        fold_voxel_window(win[i], wx0, wy0, frame, p.W, p.H, L.U[i].z, L.V[i].z, L.D[i].z, limit, tsd[2], wsum[2]);
        fold_voxel_window(win[i], wx0, wy0, frame, p.W, p.H, L.U[i].w, L.V[i].w, L.D[i].w, limit, tsd[3], wsum[3]);
    }
  }
}

template <int N>
__global__ __launch_bounds__(128, 5) void k_integrate_tiled_list(IntegrateParams p)
{
  constexpr int GS = N <= 4 ? N : (N + 1) / 2;  // sensors per stage
  constexpr int NG = (N + GS - 1) / GS;         // stages per tile (1 or 2)
  __shared__ uint2 win[GS][kWin * kWinPitch];
  __shared__ uint8_t occ_lds[kOccSide * kOccSide * kOccSide];
  const auto list = ro(p.tile_list);
  const unsigned n = *p.tile_count;
  const int q = threadIdx.x;
  const unsigned step = gridDim.x;
  const float limit = p.limit;
  auto entry_at = [&](unsigned i) { return i < n ? list[i] : 0xffffffffu; };
  // block-uniform words: the current tile's, the next tile's, and the list entry after that
  TileWords<N> w0, w1;
  load_tile_words<N>(p, entry_at(blockIdx.x), w0);
  unsigned e1 = entry_at(blockIdx.x + step);
  StageLoads<GS> L;
  OccLoads oc;
  for (unsigned i = blockIdx.x; i < n; i += step) {
    const unsigned e2 = entry_at(i + 2u * step);  // both arrive while this tile is worked on
    load_tile_words<N>(p, e1, w1);
    const unsigned tile = w0.entry & 0x7fffffffu;
    const bool whole = (w0.entry >> 31) != 0;
    float tsd[4] = {limit, limit, limit, limit};
    float wsum[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    uint32_t lo[3] = {0, 0, 0}, cnt[3] = {1, 1, 1};
    const bool cached = !whole && occ_cached(p, w0, lo, cnt);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      issue_stage<N, GS>(p, w0, g, q, L);
      if (g == 0 && !whole) issue_occ<N>(p, w0, q, oc);
      if (g == 0 && cached) {
        // the mask bytes go to LDS ahead of the window exchange of fold_stage, after a barrier of their own:
        // lanes of the previous tile may still be reading the cache
        __syncthreads();
        if ((uint32_t)q < cnt[0] * cnt[1] * cnt[2]) occ_lds[q] = oc.mask;
      }
      fold_stage<N, GS>(p, w0, g, q, L, win, limit, tsd, wsum);
    }
    if (!whole) {
      const int lz = q >> 4, ly = (q >> 1) & 7, lx0 = (q & 1) * 4;
      const int vx = (int)(tile % p.TX) * kTile + lx0, vy = (int)((tile / p.TX) % p.TY) * kTile + ly;
      const int vz = (p.tz0 + (int)(tile / (p.TX * p.TY))) * kTile + lz;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        bool any = false;
        if (cached) {  // a voxel's ranges are sub-ranges of the tile's (BrickTables::tile is their union)
          for (uint32_t bz = oc.ez & 0xffffu; bz <= (oc.ez >> 16); ++bz)
            for (uint32_t by = oc.ey & 0xffffu; by <= (oc.ey >> 16); ++by)
              for (uint32_t bx = oc.ex[j] & 0xffffu; bx <= (oc.ex[j] >> 16); ++bx)
                any |= occ_lds[((bz - lo[2]) * cnt[1] + (by - lo[1])) * cnt[0] + (bx - lo[0])] != 0;
        } else {
          any = voxel_occupied(p, vx + j, vy, vz);
        }
        if (!any) tsd[j] = -limit;
      }
    }
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f r = {tsd[0], tsd[1], tsd[2], tsd[3]};
    __builtin_nontemporal_store(r, reinterpret_cast<v4f*>(p.tsdf + (size_t)tile * kTileVoxels) + q);
    w0 = w1;
    e1 = e2;
  }
}

constexpr int kClearTiles = 256;
// Brick-skipping sweep, first half (the clear of recon_integration.cpp:246-249 for everything
// integrate will not touch).  One lane per tile: does the tile overlap an occupied brick?
// Then it goes on the work list of k_integrate_tiled_list.  Otherwise it must hold -limit --
// and if tile_state[tile] == epoch it still does from an earlier sweep (the host bumps the
// epoch whenever anything else may have written the volume or the limit changed), so a
// steady stream only rewrites the tiles the surface has just left.  Tiles that do need the
// clear are collected per block and streamed out by all 256 lanes (2 KiB each, non-temporal).
// LAZY: updateOccupiedBricks' filter rides along (rgbdr_update_occupied_bricks only noted the threshold): the
// decisions below read the counters themselves, and every lane also writes mask bytes for the sweep that follows
// and for later consumers -- one launch less per frame.
template <bool LAZY>
__global__ __launch_bounds__(256) void k_brick_clear(IntegrateParams p, unsigned ntiles)
{
  __shared__ unsigned todo[kClearTiles];
  __shared__ unsigned ntodo;
  if (threadIdx.x == 0) ntodo = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) *p.tile_count_next = 0u;  // the counter the next sweep appends to
  __syncthreads();
  const unsigned tile = blockIdx.x * kClearTiles + threadIdx.x;
  if (LAZY) {
    for (unsigned b = tile; b < (unsigned)p.num_bricks; b += gridDim.x * kClearTiles)
      p.brick_mask_out[b] = p.brick_counters[b] >= p.min_voxels ? 1 : 0;
  }
  bool any = false, whole = false, clear = false;
  if (tile < ntiles) {
    const int tx = tile % p.TX, ty = (tile / p.TX) % p.TY, tz = p.tz0 + tile / (p.TX * p.TY);
    // bricks that hold any of the tile's voxels (union of the per-coordinate brick ranges); a superset
    // of the tiles with an occupied voxel is enough here -- the sweep decides per voxel
    // (BrickTables::tile: the same lo | hi << 16 ranges per storage tile, built on the host)
    const uint32_t ex = p.tbx[tx], ey = p.tby[ty], ez = p.tbz[tz];
    const uint32_t lo[3] = {ex & 0xffffu, ey & 0xffffu, ez & 0xffffu}, hi[3] = {ex >> 16, ey >> 16, ez >> 16};
    const unsigned state = p.tile_state[tile];
    bool all = true;
    for (uint32_t bz = lo[2]; bz <= hi[2]; ++bz)
      for (uint32_t by = lo[1]; by <= hi[1]; ++by)
        for (uint32_t bx = lo[0]; bx <= hi[0]; ++bx) {
          const size_t id = ((size_t)bz * p.by + by) * p.bx + bx;
          const bool o = LAZY ? p.brick_counters[id] >= p.min_voxels : p.brick_mask[id] != 0;
          any |= o;
          all &= o;
        }
    // voxels that indices past the x / y end of the last brick alias (voxel_occupied): x < ovx or y < ovy
    any |= (p.ovx && tx * kTile < p.ovx) || (p.ovy && ty * kTile < p.ovy);
    // every brick the tile touches is occupied and every voxel of the tile lies in one of them (twx/twy/twz):
    // every voxel is occupied, the sweep skips the per-voxel test (bit 31 of the list entry)
    whole = any && all && !(p.ovx | p.ovy) && p.twx[tx] && p.twy[ty] && p.twz[tz];
    clear = !any && state != p.epoch;
    if (any)
      p.tile_state[tile] = 0u;  // about to hold integrated values
    else if (clear)
      p.tile_state[tile] = p.epoch;
  }
  // one atomic per wavefront for the list (thousands of lanes appending one entry each to one counter otherwise)
  {
    const unsigned long long m = __ballot(any);
    const int lane = threadIdx.x & 63;
    unsigned base = 0;
    if (m) {
      if (lane == __ffsll((long long)m) - 1) base = atomicAdd(p.tile_count, (unsigned)__popcll(m));
      base = __shfl(base, __ffsll((long long)m) - 1);
      if (any) p.tile_list[base + (unsigned)__popcll(m & ((1ull << lane) - 1ull))] = tile | (whole ? 0x80000000u : 0u);
    }
  }
  if (clear) todo[atomicAdd(&ntodo, 1u)] = tile;
  __syncthreads();
  const unsigned n = ntodo;
  if (n == 0) return;
  typedef float v4f __attribute__((ext_vector_type(4)));
  const float l = -p.limit;
  const v4f fill = {l, l, l, l};
  v4f* out = reinterpret_cast<v4f*>(p.tsdf);
  for (unsigned i = threadIdx.x; i < n * (kTileVoxels / 4); i += 256)
    __builtin_nontemporal_store(fill, out + (size_t)todo[i / (kTileVoxels / 4)] * (kTileVoxels / 4) + (i % (kTileVoxels / 4)));
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

// RGBDR_FLAG_SKIP_BACKGROUND, once per frame: for every origin (ox, oy) in [-1, W-1] x [-1, H-1] and the squares of
// 4, 8 and 16 (edge-clamped) texels from there, what the texels have in common, as three bounds
// ([sensor][size class][3][(H+1)][(W+1)], index (oy + 1) * (W + 1) + (ox + 1)):
//   bound 0: all background (silhouette 0, depth not NaN) -> their largest depth, else +inf
//   bound 1, 2: all surface (silhouette 1, depth not NaN) -> their smallest / largest depth, else -inf / +inf
// Squares of 8 and 16 are folded from two of the next smaller size, along x and then along y.
__global__ __launch_bounds__(256) void k_window_background(const uint2* __restrict__ frames, int W, int H,
                                                           float* __restrict__ bgmax)
{
  constexpr int T = 16 + kWin - 1;  // 31 texels per axis feed 16 origins
  __shared__ float tex[3][T][T + 1];  // rows, then reduced along x in place
  const int l = blockIdx.z;
  const uint2* frame = frames + (size_t)l * W * H;
  const int ox0 = (int)blockIdx.x * 16 - 1, oy0 = (int)blockIdx.y * 16 - 1;
  const int t = threadIdx.y * 16 + threadIdx.x;
  const float inf = __builtin_inff();
  for (int i = t; i < T * T; i += 256) {
    const int ty = i / T, tx = i - ty * T;
    const uint2 v = frame[(size_t)clampi(oy0 + ty, 0, H - 1) * W + clampi(ox0 + tx, 0, W - 1)];
    const float d = texel_depth(v);
    const bool num = d == d, bg = (v.y >> 31) != 0;
    tex[0][ty][tx] = (bg && num) ? d : inf;    // max-reduced
    tex[1][ty][tx] = (!bg && num) ? d : -inf;  // min-reduced
    tex[2][ty][tx] = (!bg && num) ? d : inf;   // max-reduced
  }
  __syncthreads();
  const int ox = ox0 + (int)threadIdx.x, oy = oy0 + (int)threadIdx.y;
  const bool live = ox <= W - 1 && oy <= H - 1;
  const size_t plane = (size_t)(W + 1) * (H + 1), o = (size_t)(oy + 1) * (W + 1) + (ox + 1);
  // squares by doubling, each level staged in LDS: rows of 4 -> squares of 4 (c4, in place of tex) -> squares of 8
  // (c8, in place of r4) -> squares of 16 from four c8
  __shared__ float r4[3][T][T + 1];
  auto red = [](int b, float x, float y) { return (b == 1) ? fminf(x, y) : fmaxf(x, y); };
  for (int i = t; i < T * (T - 3); i += 256) {
    const int ty = i / (T - 3), tx = i - ty * (T - 3);
#pragma unroll
    for (int b = 0; b < 3; ++b)
      r4[b][ty][tx] = red(b, red(b, tex[b][ty][tx], tex[b][ty][tx + 1]), red(b, tex[b][ty][tx + 2], tex[b][ty][tx + 3]));
  }
  __syncthreads();
  for (int i = t; i < (T - 3) * (T - 3); i += 256) {  // c4[y][x], x, y in [0, 27]
    const int ty = i / (T - 3), tx = i - ty * (T - 3);
#pragma unroll
    for (int b = 0; b < 3; ++b)
      tex[b][ty][tx] = red(b, red(b, r4[b][ty][tx], r4[b][ty + 1][tx]), red(b, r4[b][ty + 2][tx], r4[b][ty + 3][tx]));
  }
  __syncthreads();
  for (int i = t; i < (T - 7) * (T - 7); i += 256) {  // c8[y][x], x, y in [0, 23]
    const int ty = i / (T - 7), tx = i - ty * (T - 7);
#pragma unroll
    for (int b = 0; b < 3; ++b)
      r4[b][ty][tx] = red(b, red(b, tex[b][ty][tx], tex[b][ty][tx + 4]), red(b, tex[b][ty + 4][tx], tex[b][ty + 4][tx + 4]));
  }
  __syncthreads();
  if (!live) return;
  const int x = threadIdx.x, y = threadIdx.y;
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    const float q16 = red(b, red(b, r4[b][y][x], r4[b][y][x + 8]), red(b, r4[b][y + 8][x], r4[b][y + 8][x + 8]));
    bgmax[(((size_t)l * 3 + 0) * 3 + b) * plane + o] = tex[b][y][x];
    bgmax[(((size_t)l * 3 + 1) * 3 + b) * plane + o] = r4[b][y][x];
    bgmax[(((size_t)l * 3 + 2) * 3 + b) * plane + o] = q16;
  }
}
void launch_window_background(const uint2* frames, int W, int H, int N, float* bgmax, hipStream_t s)
{
  hipLaunchKernelGGL(k_window_background, dim3((unsigned)((W + 1 + 15) / 16), (unsigned)((H + 1 + 15) / 16), (unsigned)N),
                     dim3(16, 16), 0, s, frames, W, H, bgmax);
}

// The verdict of one (tile, sensor) pair for the current frame (kSkip*)
__device__ __forceinline__ unsigned skip_verdict(const IntegrateParams& p, size_t i, int s)
{
  const int d = p.win[i];
  const int wx0 = (int)(short)(d & 0xffff), wy0 = (int)(short)(d >> 16);
  const size_t plane = (size_t)(p.W + 1) * (p.H + 1), o = (size_t)(wy0 + 1) * (p.W + 1) + (wx0 + 1);
  const float* b = p.bgmax + ((size_t)s * 3 + (size_t)p.win_ext[i]) * 3 * plane + o;
  const float dmin = p.win_dmin[i], dmax = p.win_dmax[i];
  // every comparison is false for the "does not apply" values (dmin = -inf, dmax = +inf, bounds of a mixed window)
  if ((dmin - b[0]) >= p.limit) return kSkipCarve;
  if ((dmax - b[plane]) <= -p.limit) return kSkipFront;
  if ((dmin - b[2 * plane]) >= p.limit) return kSkipBehind;
  return kSkipNone;
}

// one byte per pair (the diagnostics of rgbdr_skipped_pairs / rgbdr_readback_skip_tables)
__global__ void k_skip_mask(IntegrateParams p, unsigned npairs, uint8_t* __restrict__ mask)
{
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < npairs) mask[i] = (uint8_t)skip_verdict(p, i, (int)(i % (unsigned)p.N));
}

// First half of the RGBDR_FLAG_SKIP_BACKGROUND sweep (the role k_brick_clear has for bricks): one lane per tile
// takes the verdicts of its N sensors.  If every sensor has one, the tile's 512 voxels all end as the same value --
// over tsd = limit the first carve or in-front verdict makes -limit and nothing after it changes that; hidden from
// every sensor: +limit -- and the tile is filled here, unless tile_state says it has held -limit since a sweep of
// this epoch (the bookkeeping of the brick sweep and of RGBDR_FLAG_ELIDE_STORES; this sweep always keeps it).
// Otherwise tile | verdicts << 32 goes on the list of k_integrate_tiled_listed.
constexpr int kClassifyTiles = 256;
__global__ __launch_bounds__(256) void k_skip_classify(IntegrateParams p, unsigned ntiles)
{
  __shared__ unsigned todo[kClassifyTiles];  // tile | (value is +limit) << 31
  __shared__ unsigned ntodo;
  if (threadIdx.x == 0) ntodo = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) *p.skip_count_next = 0u;  // the counter the next sweep appends to
  __syncthreads();
  const unsigned tile = blockIdx.x * kClassifyTiles + threadIdx.x;
  bool listed = false, fill = false, positive = false;
  unsigned actions = 0u;
  if (tile < ntiles) {
    bool all = true, negative = false;
    for (int s = 0; s < p.N; ++s) {
      const unsigned a = skip_verdict(p, (size_t)tile * p.N + s, s);
      actions |= a << (2 * s);
      all = all && a != kSkipNone;
      negative = negative || a == kSkipCarve || a == kSkipFront;
    }
    listed = !all;
    if (all) {
      positive = !negative;
      fill = !(negative && p.tile_state[tile] == p.epoch);
      p.tile_state[tile] = negative ? p.epoch : 0u;
    } else {
      p.tile_state[tile] = 0u;  // about to hold integrated values
    }
  }
  {
    const unsigned long long m = __ballot(listed);
    const int lane = threadIdx.x & 63;
    unsigned base = 0;
    if (m) {
      if (lane == __ffsll((long long)m) - 1) base = atomicAdd(p.skip_count, (unsigned)__popcll(m));
      base = __shfl(base, __ffsll((long long)m) - 1);
      if (listed) p.skip_list[base + (unsigned)__popcll(m & ((1ull << lane) - 1ull))] = (unsigned long long)tile | ((unsigned long long)actions << 32);
    }
  }
  if (fill) todo[atomicAdd(&ntodo, 1u)] = tile | (positive ? 0x80000000u : 0u);
  __syncthreads();
  const unsigned n = ntodo;
  if (n == 0) return;
  typedef float v4f __attribute__((ext_vector_type(4)));
  v4f* out = reinterpret_cast<v4f*>(p.tsdf);
  for (unsigned i = threadIdx.x; i < n * (kTileVoxels / 4); i += 256) {
    const unsigned e = todo[i / (kTileVoxels / 4)];
    const float l = (e >> 31) ? p.limit : -p.limit;
    const v4f fillv = {l, l, l, l};
    __builtin_nontemporal_store(fillv, out + (size_t)(e & 0x7fffffffu) * (kTileVoxels / 4) + (i % (kTileVoxels / 4)));
  }
}

// Second half: one block per listed tile (the host sizes the grid from the previous frame's list length; blocks
// stride over the list, so any grid is correct).  Verdicts come with the list entry: no load in front of the
// LUT loads but the entry itself.
template <int N>
__global__ __launch_bounds__(128, N <= 7 ? 5 : 4) void k_integrate_tiled_listed(IntegrateParams p)
{
  constexpr int G1 = N <= 4 ? N : (N + 1) / 2;
  constexpr int G2 = N - G1;
  __shared__ uint2 win[G1][kWin * kWinPitch];
  const unsigned n = *ro(p.skip_count);
  const int q = threadIdx.x;
  const float limit = p.limit;
  for (unsigned i = blockIdx.x; i < n; i += gridDim.x) {
    const unsigned long long e = ro(p.skip_list)[i];
    const unsigned tile = (unsigned)e, actions = (unsigned)(e >> 32);
    float tsd[4] = {limit, limit, limit, limit};
    float wsum[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    integrate_group<G1, true, true>(p, tile, q, 0, N, win, i != blockIdx.x, limit, tsd, wsum, actions);
    if (G2 > 0) integrate_group<(G2 > 0 ? G2 : 1), true, true>(p, tile, q, G1, N, win, true, limit, tsd, wsum, actions);
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f r = {tsd[0], tsd[1], tsd[2], tsd[3]};
    __builtin_nontemporal_store(r, reinterpret_cast<v4f*>(p.tsdf + (size_t)tile * kTileVoxels) + q);
  }
}

void launch_skip_mask(const IntegrateParams& p, unsigned npairs, uint8_t* mask, hipStream_t s)
{
  hipLaunchKernelGGL(k_skip_mask, dim3((npairs + 255) / 256), dim3(256), 0, s, p, npairs, mask);
}
template <int N>
static void launch_listed_n(const IntegrateParams& p, unsigned blocks, hipStream_t s)
{
  hipLaunchKernelGGL((k_integrate_tiled_listed<N>), dim3(blocks), dim3(128), 0, s, p);
}
// the background-skip sweep: classifier + one block per listed tile (`blocks`: the host's estimate of the list length)
void launch_skip_sweep(const IntegrateParams& p, unsigned blocks, hipStream_t s)
{
  const unsigned ntiles = (unsigned)p.TX * p.TY * p.ntz;
  hipLaunchKernelGGL(k_skip_classify, dim3((ntiles + kClassifyTiles - 1) / kClassifyTiles), dim3(256), 0, s, p, ntiles);
  switch (p.N) {
    case 1: launch_listed_n<1>(p, blocks, s); break;
    case 2: launch_listed_n<2>(p, blocks, s); break;
    case 3: launch_listed_n<3>(p, blocks, s); break;
    case 4: launch_listed_n<4>(p, blocks, s); break;
    case 5: launch_listed_n<5>(p, blocks, s); break;
    case 6: launch_listed_n<6>(p, blocks, s); break;
    case 7: launch_listed_n<7>(p, blocks, s); break;
    default: launch_listed_n<8>(p, blocks, s); break;
  }
}
// number of non-zero mask bytes (diagnostic, on demand: thousands of atomics on one word cost more than the mask itself)
__global__ __launch_bounds__(1024) void k_count_bytes(const uint8_t* __restrict__ mask, unsigned n, unsigned* __restrict__ count)
{
  __shared__ unsigned part[16];
  unsigned c = 0;
  for (unsigned i = blockIdx.x * 1024u + threadIdx.x; i < n; i += gridDim.x * 1024u) c += mask[i] != 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned t = 0;
    for (int w = 0; w < 16; ++w) t += part[w];
    atomicAdd(count, t);
  }
}
void launch_count_bytes(const uint8_t* mask, unsigned n, unsigned* count, hipStream_t s)
{
  hipLaunchKernelGGL(k_count_bytes, dim3(64), dim3(1024), 0, s, mask, n, count);
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

template <int N>
static void launch_tiled_n(const IntegrateParams& p, unsigned ntiles, hipStream_t s)
{
  if (p.use_bricks) {
    if (p.brick_counters)
      hipLaunchKernelGGL(k_brick_clear<true>, dim3((ntiles + kClearTiles - 1) / kClearTiles), dim3(256), 0, s, p, ntiles);
    else
      hipLaunchKernelGGL(k_brick_clear<false>, dim3((ntiles + kClearTiles - 1) / kClearTiles), dim3(256), 0, s, p, ntiles);
    const unsigned blocks = ntiles < 2560u ? ntiles : 2560u;  // 10 resident blocks (5 wavefronts per SIMD) on each of the 256 CUs
    hipLaunchKernelGGL((k_integrate_tiled_list<N>), dim3(blocks), dim3(128), 0, s, p);
    return;
  }
  // developer A/B knob: RGBDR_INTEGRATE_GROUP=2 folds 3 or 4 sensors in two groups
  static const int maxg = getenv("RGBDR_INTEGRATE_GROUP") ? atoi(getenv("RGBDR_INTEGRATE_GROUP")) : 4;
  if (maxg == 2 && (N == 3 || N == 4)) {
    hipLaunchKernelGGL((k_integrate_tiled<N, 2>), dim3(ntiles), dim3(128), 0, s, p);
    return;
  }
  // developer A/B knob: RGBDR_NT=0 uses temporal loads/stores for the LUT / TSDF streams
  if (getenv("RGBDR_NT") && !atoi(getenv("RGBDR_NT")) && N == 4) {
    hipLaunchKernelGGL((k_integrate_tiled<N, 4, false>), dim3(ntiles), dim3(128), 0, s, p);
    return;
  }
  if (p.elide_stores)
    hipLaunchKernelGGL((k_integrate_tiled<N, 4, true, true>), dim3(ntiles), dim3(128), 0, s, p);
  else if (p.stage_lo || p.stage_hi)
    hipLaunchKernelGGL((k_integrate_tiled<N, 4, true, false, true>), dim3(ntiles), dim3(128), 0, s, p);
  else
    hipLaunchKernelGGL((k_integrate_tiled<N>), dim3(ntiles), dim3(128), 0, s, p);
}

// true when launch_integrate's kernel itself fills the halo staging buffers (plain full sweep of a
// 1:1 / resampled LUT); for every other sweep the caller copies the layers afterwards
bool integrate_stages_halo(const IntegrateParams& p, bool one_to_one)
{
  if (!one_to_one || p.use_bricks || p.elide_stores || p.skip_background) return false;
  static const bool knobs = getenv("RGBDR_INTEGRATE_GROUP") || getenv("RGBDR_NT");
  return !knobs;
}

void launch_integrate(const IntegrateParams& p_in, bool one_to_one, hipStream_t s)
{
  IntegrateParams p = p_in;
  const unsigned ntiles = (unsigned)p.TX * p.TY * p.ntz;
  // full sweep: chunk = one x-row of tiles when the grid divides evenly (measured
  // 3-8 % faster than identity / one contiguous run per XCD); brick-skipping sweep:
  // identity (most blocks only clear their tile; chunked order measured 40 % slower).
  // RGBDR_TILE_CHUNK overrides (developer knob: 0 = identity, -1 = ntiles/8)
  unsigned chunk = p.use_bricks ? 0u : (unsigned)p.TX;
  if (const char* e = getenv("RGBDR_TILE_CHUNK")) chunk = atoi(e) < 0 ? ntiles / 8 : (unsigned)atoi(e);
  p.order_chunk = (chunk && ntiles % (8 * chunk) == 0) ? chunk : 0;
  if (!one_to_one) {
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
