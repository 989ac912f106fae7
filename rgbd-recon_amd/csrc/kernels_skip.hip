// kernels_skip.hip -- RGBDR_FLAG_SKIP_BACKGROUND: the full sweep with verdicts instead of LUT planes (integrate_fold.cuh:
// integrate_group<..., SKIP>).  Per frame: k_window_background (what the squares of 4 / 8 / 16 texels at
// every window origin have in common), k_skip_classify (verdicts per tile; constant tiles filled, the others listed),
// k_integrate_tiled_listed (one block per listed tile); k_skip_mask / k_count_bytes are the diagnostics.
#include <hip/hip_runtime.h>

#include "integrate_fold.cuh"

namespace rgbdr {

// RGBDR_FLAG_SKIP_BACKGROUND, once per frame: for every origin (ox, oy) in [-1, W-1] x [-1, H-1] and the squares of
// 4, 8 and 16 (edge-clamped) texels from there, what the texels have in common, as three bounds
// ([sensor][size class][3][(H+1)][(W+1)], index (oy + 1) * (W + 1) + (ox + 1)):
//   bound 0: all background (silhouette 0, depth not NaN) -> their largest depth, else +inf
//   bound 1, 2: all surface (silhouette 1, depth not NaN) -> their smallest / largest depth, else -inf / +inf
// Squares of 8 and 16 are folded from two of the next smaller size, along x and then along y.
// One block serves O x O origins from (O + 15)^2 texels: with O = 32 a texel is staged 2.2 times (O = 16: 3.75).
// Every stage walks its array with a linear index whose row width IS the LDS pitch P, so the 32 lanes of a
// ds_read_b32 group touch 32 consecutive words: no bank conflicts and no division by an odd row width (round 3:
// index rows of 28 / 24 over a pitch of 32 -- 49 % of the kernel's LDS cycles were conflicts).  Entries whose window
// runs off the row or the array are folded from whatever lies there and never read by an origin of the block.
// Round 4: 21.9 -> 15.9 us at 4 x 512 x 424 (O = 16 with the linear index: 19.8; four entries per lane with 16-byte
// LDS accesses: 16.3 -- LDS is no longer what the kernel waits for; the three bounds of an origin in one 16-byte entry
// instead of three planes: the same here and +1.8 us in the classifier; profiles/r04_notes/experiments.md).
template <int O>
__global__ __launch_bounds__(O * O) void k_window_background(const uint2* __restrict__ frames, int W, int H,
                                                             float* __restrict__ bgmax)
{
  constexpr int T = O + kWin - 1;  // texels per axis feeding O origins (47)
  constexpr int P = T + 1;         // pitch = index row width (48)
  constexpr int NT = O * O;
  __shared__ float bufa[3][T * P + 4], bufb[3][T * P + 4];  // + 4: the tail of the last row reads past it
  const int l = blockIdx.z;
  const uint2* frame = frames + (size_t)l * W * H;
  const int ox0 = (int)blockIdx.x * O - 1, oy0 = (int)blockIdx.y * O - 1;
  const int t = threadIdx.y * O + threadIdx.x;
  const float inf = __builtin_inff();
  for (int i = t; i < T * P; i += NT) {
    const int ty = i / P, tx = i - ty * P;
    const uint2 v = frame[(size_t)clampi(oy0 + ty, 0, H - 1) * W + clampi(ox0 + tx, 0, W - 1)];
    const float d = texel_depth(v);
    const bool num = d == d, bg = (v.y >> 31) != 0;
    bufa[0][i] = (bg && num) ? d : inf;    // max-reduced
    bufa[1][i] = (!bg && num) ? d : -inf;  // min-reduced
    bufa[2][i] = (!bg && num) ? d : inf;   // max-reduced
  }
  if (t < 12) bufa[t >> 2][T * P + (t & 3)] = 0.0f;
  __syncthreads();
  auto red = [](int b, float x, float y) { return (b == 1) ? fminf(x, y) : fmaxf(x, y); };
  // squares by doubling, each level staged in LDS: rows of 4 (bufb) -> squares of 4 (c4, bufa) -> squares of 8 (c8, bufb)
  // -> squares of 16 from four c8
  for (int i = t; i < T * P; i += NT) {
#pragma unroll
    for (int b = 0; b < 3; ++b)
      bufb[b][i] = red(b, red(b, bufa[b][i], bufa[b][i + 1]), red(b, bufa[b][i + 2], bufa[b][i + 3]));
  }
  __syncthreads();
  for (int i = t; i < (T - 3) * P; i += NT) {  // c4[y][x], x, y in [0, T - 4]
#pragma unroll
    for (int b = 0; b < 3; ++b)
      bufa[b][i] = red(b, red(b, bufb[b][i], bufb[b][i + P]), red(b, bufb[b][i + 2 * P], bufb[b][i + 3 * P]));
  }
  __syncthreads();
  for (int i = t; i < (T - 7) * P; i += NT) {  // c8[y][x], x, y in [0, T - 8]
#pragma unroll
    for (int b = 0; b < 3; ++b)
      bufb[b][i] = red(b, red(b, bufa[b][i], bufa[b][i + 4]), red(b, bufa[b][i + 4 * P], bufa[b][i + 4 * P + 4]));
  }
  __syncthreads();
  const int ox = ox0 + (int)threadIdx.x, oy = oy0 + (int)threadIdx.y;
  if (ox > W - 1 || oy > H - 1) return;
  const size_t plane = (size_t)(W + 1) * (H + 1), o = (size_t)(oy + 1) * (W + 1) + (ox + 1);
  const int c = (int)threadIdx.y * P + (int)threadIdx.x;
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    const float q16 = red(b, red(b, bufb[b][c], bufb[b][c + 8]), red(b, bufb[b][c + 8 * P], bufb[b][c + 8 * P + 8]));
#ifdef RGBDR_WB_NO_STORE
    if (q16 + bufa[b][c] + bufb[b][c] != 12345.678f) continue;  // diagnostic build: everything but the stores
#endif
    bgmax[(((size_t)l * 3 + 0) * 3 + b) * plane + o] = bufa[b][c];
    bgmax[(((size_t)l * 3 + 1) * 3 + b) * plane + o] = bufb[b][c];
    bgmax[(((size_t)l * 3 + 2) * 3 + b) * plane + o] = q16;
  }
}
void launch_window_background(const uint2* frames, int W, int H, int N, float* bgmax, hipStream_t s)
{
  constexpr int O = 32;
  hipLaunchKernelGGL((k_window_background<O>), dim3((unsigned)((W + 1 + O - 1) / O), (unsigned)((H + 1 + O - 1) / O), (unsigned)N),
                     dim3(O, O), 0, s, frames, W, H, bgmax);
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

// First half of the RGBDR_FLAG_SKIP_BACKGROUND sweep (the role k_brick_clear has for bricks): the verdicts of a
// tile's N sensors for the current frame.  If every sensor has one, the tile's 512 voxels all end as the same value --
// over tsd = limit the first carve or in-front verdict makes -limit and nothing after it changes that; hidden from
// every sensor: +limit -- and the tile is filled here, unless tile_state says it has held -limit since a sweep of
// this epoch (the bookkeeping of the brick sweep and of RGBDR_FLAG_ELIDE_STORES; this sweep always keeps it).
// Otherwise {tile, verdicts, window origins} goes on the list of k_integrate_tiled_listed.
// One lane per tile, 1024 tiles per block.  The block's 1024 N pairs are taken N per lane -- the four words of a pair
// are read coalesced, the three bounds behind the window origin are one dependent round trip, all N of a lane in
// flight at once -- then every lane combines its tile's verdicts from LDS, and the block appends its listed tiles
// with ONE atomic on the list counter.  That counter is what the kernel's time follows -- a returning atomic on one
// address costs about 10 ns (round 2: one per wavefront of 64 tiles, 4096 of them at 512^3, 48 us; round 3: one per 256
// tiles, 20.3 us, and that kernel with 128 / 64 tiles per block 28.4 / 48.7 us) -- until the block's own chain of dependent round trips
// is what is left: one per 1024 tiles (256 blocks at 512^3, one per CU) 18.5 us, one per 512 tiles in this form
// 20.7 us (profiles/r04_notes/experiments.md).
constexpr int kClassifyThreads = 1024;
template <int N>
__global__ __launch_bounds__(kClassifyThreads) void k_skip_classify(IntegrateParams p, unsigned ntiles)
{
  __shared__ unsigned char verdict[N * kClassifyThreads];  // [tile in block][sensor]
  __shared__ int origin[N * kClassifyThreads];             // the pair's window origin word (goes into the list entry)
  __shared__ unsigned todo[kClassifyThreads];              // tile | (value is +limit) << 31
  __shared__ unsigned ntodo, wave_listed[kClassifyThreads / 64], list_base;
  if (threadIdx.x == 0) ntodo = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) *p.skip_count_next = 0u;  // the counter the next sweep appends to
  const unsigned tile0 = blockIdx.x * (unsigned)kClassifyThreads;
  const size_t pair0 = (size_t)tile0 * N, pair_end = (size_t)ntiles * N;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const unsigned j = (unsigned)k * kClassifyThreads + threadIdx.x;  // pair of the block; pair0 is a multiple of N
    if (pair0 + j < pair_end) {
      origin[j] = p.win[pair0 + j];
      verdict[j] = (unsigned char)skip_verdict(p, pair0 + j, (int)(j % (unsigned)N));
    }
  }
  __syncthreads();
  const unsigned tile = tile0 + threadIdx.x;
  bool listed = false, fill = false, positive = false;
  unsigned actions = 0u;
  if (tile < ntiles) {
    bool all = true, negative = false;
#pragma unroll
    for (unsigned s = 0; s < (unsigned)N; ++s) {
      const unsigned a = verdict[threadIdx.x * N + s];
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
  const unsigned long long m = __ballot(listed);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) wave_listed[wave] = (unsigned)__popcll(m);
  if (fill) todo[atomicAdd(&ntodo, 1u)] = tile | (positive ? 0x80000000u : 0u);
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned total = 0;
    for (int w = 0; w < kClassifyThreads / 64; ++w) {
      const unsigned c = wave_listed[w];
      wave_listed[w] = total;  // exclusive prefix
      total += c;
    }
    list_base = total ? atomicAdd(p.skip_count, total) : 0u;
  }
  __syncthreads();
  if (listed) {  // entry: tile, verdicts, the N window origins (so that the sweep's window loads wait for one load, not two)
    unsigned* e = p.skip_list + (size_t)(list_base + wave_listed[wave] + (unsigned)__popcll(m & ((1ull << lane) - 1ull))) * (2 + N);
    e[0] = tile;
    e[1] = actions;
#pragma unroll
    for (unsigned s = 0; s < (unsigned)N; ++s) e[2 + s] = (unsigned)origin[threadIdx.x * N + s];
  }
  const unsigned n = ntodo;
  if (n == 0) return;
  typedef float v4f __attribute__((ext_vector_type(4)));
  v4f* out = reinterpret_cast<v4f*>(p.tsdf);
  for (unsigned i = threadIdx.x; i < n * (kTileVoxels / 4); i += kClassifyThreads) {
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
  if (blockIdx.x == 0 && q == 0 && p.skip_count_host) {  // the host sizes the next sweep's grid from this (page-locked word)
    *p.skip_count_host = n;
    __threadfence_system();
  }
  for (unsigned i = blockIdx.x; i < n; i += gridDim.x) {
    const auto e = ro(p.skip_list) + (size_t)i * (2 + N);
    const unsigned tile = e[0], actions = e[1];
    int origins[N];
#pragma unroll
    for (int s = 0; s < N; ++s) origins[s] = (int)e[2 + s];
    float tsd[4] = {limit, limit, limit, limit};
    float wsum[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    integrate_group<G1, true, true>(p, tile, q, 0, N, win, i != blockIdx.x, limit, tsd, wsum, actions, origins);
    if (G2 > 0) integrate_group<(G2 > 0 ? G2 : 1), true, true>(p, tile, q, G1, N, win, true, limit, tsd, wsum, actions, origins);
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
static void launch_skip_sweep_n(const IntegrateParams& p, unsigned ntiles, unsigned blocks, hipStream_t s)
{
  hipLaunchKernelGGL((k_skip_classify<N>), dim3((ntiles + kClassifyThreads - 1) / kClassifyThreads), dim3(kClassifyThreads), 0, s, p, ntiles);
  hipLaunchKernelGGL((k_integrate_tiled_listed<N>), dim3(blocks), dim3(128), 0, s, p);
}
// the background-skip sweep: classifier + one block per listed tile (`blocks`: the host's estimate of the list length)
void launch_skip_sweep(const IntegrateParams& p, unsigned blocks, hipStream_t s)
{
  const unsigned ntiles = (unsigned)p.TX * p.TY * p.ntz;
  switch (p.N) {
    case 1: launch_skip_sweep_n<1>(p, ntiles, blocks, s); break;
    case 2: launch_skip_sweep_n<2>(p, ntiles, blocks, s); break;
    case 3: launch_skip_sweep_n<3>(p, ntiles, blocks, s); break;
    case 4: launch_skip_sweep_n<4>(p, ntiles, blocks, s); break;
    case 5: launch_skip_sweep_n<5>(p, ntiles, blocks, s); break;
    case 6: launch_skip_sweep_n<6>(p, ntiles, blocks, s); break;
    case 7: launch_skip_sweep_n<7>(p, ntiles, blocks, s); break;
    default: launch_skip_sweep_n<8>(p, ntiles, blocks, s); break;
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

}  // namespace rgbdr
