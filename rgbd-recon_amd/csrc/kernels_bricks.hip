// kernels_bricks.hip -- the brick-skipping sweep (the reference's default, m_use_bricks):
// ReconIntegration::integrate with the per-brick index lists (recon_integration.cpp:243-270, 361-388) as
// k_brick_clear (which tiles touch an occupied brick; the others are cleared if they are not clear already) +
// k_integrate_tiled_list (persistent blocks walk the list).  DESIGN.md 4.7.
#include <hip/hip_runtime.h>

#include "integrate_fold.cuh"

namespace rgbdr {

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
// kClearTiles lanes (= tiles) per block: 1024 for large grids (one list atomic per 1024 tiles), 256 for small ones
// (a 200 x 221 x 200 grid has 17 500 tiles: 18 blocks of 1024 would leave most of the machine idle).
template <bool LAZY, int kClearTiles>
__global__ __launch_bounds__(kClearTiles) void k_brick_clear(IntegrateParams p, unsigned ntiles)
{
  __shared__ unsigned todo[kClearTiles];
  __shared__ unsigned ntodo, wave_listed[kClearTiles / 64], list_base;
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
  // one atomic per BLOCK for the list: same-address atomics serialise in the L2 (one per wavefront was measured as
  // most of the classifier of the background-skip sweep, kernels_skip.hip)
  const unsigned long long m = __ballot(any);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) wave_listed[wave] = (unsigned)__popcll(m);
  if (clear) todo[atomicAdd(&ntodo, 1u)] = tile;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned total = 0;
    for (int w = 0; w < kClearTiles / 64; ++w) {
      const unsigned c = wave_listed[w];
      wave_listed[w] = total;  // exclusive prefix
      total += c;
    }
    list_base = total ? atomicAdd(p.tile_count, total) : 0u;
  }
  __syncthreads();
  if (any)
    p.tile_list[list_base + wave_listed[wave] + (unsigned)__popcll(m & ((1ull << lane) - 1ull))] = tile | (whole ? 0x80000000u : 0u);
  const unsigned n = ntodo;
  if (n == 0) return;
  typedef float v4f __attribute__((ext_vector_type(4)));
  const float l = -p.limit;
  const v4f fill = {l, l, l, l};
  v4f* out = reinterpret_cast<v4f*>(p.tsdf);
  for (unsigned i = threadIdx.x; i < n * (kTileVoxels / 4); i += kClearTiles)
    __builtin_nontemporal_store(fill, out + (size_t)todo[i / (kTileVoxels / 4)] * (kTileVoxels / 4) + (i % (kTileVoxels / 4)));
}

template <int N>
static void launch_list_n(const IntegrateParams& p, unsigned blocks, hipStream_t s)
{
  hipLaunchKernelGGL((k_integrate_tiled_list<N>), dim3(blocks), dim3(128), 0, s, p);
}
void launch_brick_sweep(const IntegrateParams& p, unsigned ntiles, hipStream_t s)
{
  const bool big = ntiles > 65536u;
  const unsigned t = big ? 1024u : 256u, grid = (ntiles + t - 1) / t;
  if (p.brick_counters) {
    if (big) hipLaunchKernelGGL((k_brick_clear<true, 1024>), dim3(grid), dim3(t), 0, s, p, ntiles);
    else hipLaunchKernelGGL((k_brick_clear<true, 256>), dim3(grid), dim3(t), 0, s, p, ntiles);
  } else {
    if (big) hipLaunchKernelGGL((k_brick_clear<false, 1024>), dim3(grid), dim3(t), 0, s, p, ntiles);
    else hipLaunchKernelGGL((k_brick_clear<false, 256>), dim3(grid), dim3(t), 0, s, p, ntiles);
  }
  const unsigned blocks = ntiles < 2560u ? ntiles : 2560u;  // 10 resident blocks (5 wavefronts per SIMD) on each of the 256 CUs
  switch (p.N) {
    case 1: launch_list_n<1>(p, blocks, s); break;
    case 2: launch_list_n<2>(p, blocks, s); break;
    case 3: launch_list_n<3>(p, blocks, s); break;
    case 4: launch_list_n<4>(p, blocks, s); break;
    case 5: launch_list_n<5>(p, blocks, s); break;
    case 6: launch_list_n<6>(p, blocks, s); break;
    case 7: launch_list_n<7>(p, blocks, s); break;
    default: launch_list_n<8>(p, blocks, s); break;
  }
}

}  // namespace rgbdr
