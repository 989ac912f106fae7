// api_calib.cpp -- calibration volumes behind CalibVolumes (framework/calibration/CalibVolumes.cpp:22-159):
// forward LUT upload, the grid-layout inverse-LUT arena, inverse-LUT generation on the device.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <algorithm>
#include <utility>
#include <vector>

#include "context.hpp"

using namespace rgbdr;

// A calibration volume is a 3-D texture in the reference (GL_MAX_3D_TEXTURE_SIZE: 2048 ... 16384); an inverse LUT may be as
// fine as the voxel grid (kMaxRes).  Bounding every axis keeps the products below far from 2^64 and the int coordinates of
// the kernels exact.
static const char* lut_res_error(const uint32_t res[3])
{
  for (int a = 0; a < 3; ++a) {
    if (res[a] < 1) return "empty calibration volume";
    if (res[a] > (uint32_t)kMaxRes) return "calibration volume with more than 32768 cells along an axis";
  }
  return nullptr;
}

extern "C" {
// ---------------------------------------------------------------------------
int rgbdr_set_calibration(rgbdr_ctx* ctx, int sensor, const rgbdr_lut* xyz, const rgbdr_lut* uv)
try {
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (sensor < 0 || sensor >= nsens(ctx)) return ctx->fail(RGBDR_ERR_OUT_OF_RANGE, "sensor index out of range");
  if (!xyz || !uv || !xyz->data || !uv->data) return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, "null calibration volume");
  if (const char* e = lut_res_error(xyz->res)) return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, e);
  if (const char* e = lut_res_error(uv->res)) return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, e);
  if (!(xyz->depth_limits[1] > xyz->depth_limits[0]))
    return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, "cv_xyz depth limits must satisfy max > min");
  HIPCHK(hipSetDevice(ctx->device));
  const size_t nx = (size_t)xyz->res[0] * xyz->res[1] * xyz->res[2];
  const size_t nu = (size_t)uv->res[0] * uv->res[1] * uv->res[2];
  (void)hipFree(ctx->d_cv_xyz[sensor]);
  (void)hipFree(ctx->d_cv_uv[sensor]);
  ctx->d_cv_xyz[sensor] = nullptr;
  ctx->d_cv_uv[sensor] = nullptr;
  ctx->have_calib[sensor] = false;
  DevScratch tmp;
  HIPCHK(hipMalloc(&tmp.p, nx * 12));
  HIPCHK(hipMalloc((void**)&ctx->d_cv_xyz[sensor], nx * 16));
  HIPCHK(hipMalloc((void**)&ctx->d_cv_uv[sensor], nu * 8));
  HIPCHK(hipMemcpyAsync(tmp.p, xyz->data, nx * 12, hipMemcpyHostToDevice, ctx->stream));
  launch_repack_xyz(tmp.as<float>(), ctx->d_cv_xyz[sensor], nx, ctx->stream);
  LAUNCHCHK("repack_xyz");
  HIPCHK(hipMemcpyAsync(ctx->d_cv_uv[sensor], uv->data, nu * 8, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  for (int a = 0; a < 3; ++a) {
    ctx->xyz_res[sensor][a] = xyz->res[a];
    ctx->uv_res[sensor][a] = uv->res[a];
  }
  ctx->min_ds[sensor] = xyz->depth_limits[0];
  ctx->max_ds[sensor] = xyz->depth_limits[1];
  camera_position((const float*)xyz->data, xyz->res, ctx->cam_pos[sensor]);
  frustum_planes((const float*)xyz->data, xyz->res, ctx->planes[sensor]);
  ctx->lattice_folded[sensor] = lattice_folds((const float*)xyz->data, xyz->res);
  {  // the lookups of pre_depth.fs that depend on the pixel only (kernels_pre.hip k_pre_cache)
    PreParams p{};
    p.W = ctx->cfg.depth_w;
    p.H = ctx->cfg.depth_h;
    for (int a = 0; a < 3; ++a) {
      p.bbox_min[a] = ctx->cfg.bbox_min[a];
      p.bbox_max[a] = ctx->cfg.bbox_max[a];
      p.xyz_res[sensor][a] = (int)xyz->res[a];
      p.uv_res[sensor][a] = (int)uv->res[a];
    }
    p.cv_xyz[sensor] = ctx->d_cv_xyz[sensor];
    p.cv_uv[sensor] = ctx->d_cv_uv[sensor];
    p.cc_far = ctx->d_cc_far;
    p.box_flags = ctx->d_box_flags;
    launch_pre_cache(p, sensor, ctx->stream);
    LAUNCHCHK("pre_cache");
    HIPCHK(hipStreamSynchronize(ctx->stream));
  }
  ctx->have_calib[sensor] = true;
  return RGBDR_OK;
}
RGBDR_CONTAIN(ctx)

}  // extern "C"
LutExtent rgbdr::lut_extent(const rgbdr_ctx* ctx)
{
  const rgbdr_geometry& g = ctx->geo;
  LutExtent e;
  e.t0 = g.slab_tile_z0 - ctx->halo < 0 ? 0 : g.slab_tile_z0 - ctx->halo;
  e.t1 = g.slab_tile_z1 + ctx->halo > g.tiles[2] ? g.tiles[2] : g.slab_tile_z1 + ctx->halo;
  e.vz0 = e.t0 * kTile;
  e.vz1 = e.t1 * kTile > g.res_volume[2] ? g.res_volume[2] : e.t1 * kTile;
  const ptrdiff_t layer = (ptrdiff_t)g.tiles[0] * g.tiles[1] * ctx->cfg.num_sensors * 3 * kTileVoxels;
  e.dst = ctx->d_lut_tiled + (ptrdiff_t)(e.t0 - g.slab_tile_z0) * layer;
  return e;
}
extern "C" {

}  // extern "C"
// ---------------------------------------------------------------------------
// An arena assembled from the FASTEST physical chunks (HIP's virtual-memory API).  Physical memory streams at two levels
// per 1-GiB chunk, in contiguous runs that differ per box (profiles/r05_vmm_chunk_map.txt); where no plain candidate
// placement reaches the fast level -- or there is no room to hold several -- a pool of chunks is created, each is timed
// with the sweep's pair of streams, the fastest are mapped into one reserved range and the rest released.  Such a range
// streams 0-3 % below an all-fast plain allocation and up to 10 % above a slow one, so it is only tried where no plain
// candidate was fast, and only replaces a plain arena it beats.  Any failure on the way leaves everything as it was (returns false).
struct ChunkPool {
  void* va = nullptr;
  size_t chunk = 0;
  std::vector<hipMemGenericAllocationHandle_t> handles;
};
namespace rgbdr {
void free_lut_arena(rgbdr_ctx* c)
{
  if (c->lut_vmm_va) {
    for (size_t i = 0; i < c->lut_vmm_handles.size(); ++i) {
      (void)hipMemUnmap((char*)c->lut_vmm_va + i * c->lut_vmm_chunk, c->lut_vmm_chunk);
      (void)hipMemRelease(c->lut_vmm_handles[i]);
    }
    (void)hipMemAddressFree(c->lut_vmm_va, c->lut_vmm_handles.size() * c->lut_vmm_chunk);
    (void)hipGetLastError();
    c->lut_vmm_va = nullptr;
    c->lut_vmm_handles.clear();
  } else {
    (void)hipFree(c->d_lut_tiled_base);
  }
  c->d_lut_tiled_base = nullptr;
}
}  // namespace rgbdr
// returns the base of a range of `bytes` made of the fastest chunks and its replay time, or nullptr
static float* build_chunk_arena(rgbdr_ctx* ctx, size_t bytes, size_t head_floats, size_t ntiles, float* sink, size_t chunk,
                                float* ms_out, std::vector<hipMemGenericAllocationHandle_t>& kept)
{
  const rgbdr_geometry& g = ctx->geo;
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = ctx->device;
  size_t gran = 0;
  if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || gran == 0 || chunk % gran) {
    (void)hipGetLastError();
    return nullptr;
  }
  const size_t need = (bytes + chunk - 1) / chunk;
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return nullptr;
  const size_t spare = (size_t)4 << 30;
  if (free_b < spare + (need + 2) * chunk) return nullptr;
  size_t pool_n = (free_b - spare) / chunk;
  // the pool: the arena plus as many chunks again, but between 8 and 24 -- a 26-GiB arena used to take 52 + GiB of
  // transient memory here, enough to fail another context of the process that allocated in that window
  const size_t extra = need < 8 ? 8 : (need > 24 ? 24 : need);
  const size_t want = need + extra;
  if (pool_n > want) pool_n = want;
  // chunk-level replay: per tile N x 3 planes read, one stored (the volume must hold a chunk's worth of tiles)
  const size_t tile_bytes = (size_t)nsens(ctx) * 3 * kTileVoxels * sizeof(float);
  size_t chunk_tiles = chunk / tile_bytes;
  if (chunk_tiles > ntiles) chunk_tiles = ntiles;
  if (chunk_tiles == 0) return nullptr;
  ChunkPool pool;
  pool.chunk = chunk;
  if (hipMemAddressReserve(&pool.va, pool_n * chunk, 0, nullptr, 0) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  for (size_t i = 0; i < pool_n; ++i) {
    hipMemGenericAllocationHandle_t h;
    if (hipMemCreate(&h, chunk, &prop, 0) != hipSuccess) break;
    if (hipMemMap((char*)pool.va + i * chunk, chunk, 0, h, 0) != hipSuccess) {
      (void)hipMemRelease(h);
      break;
    }
    pool.handles.push_back(h);
  }
  (void)hipGetLastError();
  const size_t got = pool.handles.size();
  if (got < need || hipMemSetAccess(pool.va, got * chunk, &acc, 1) != hipSuccess) {
    // (an address range can only be freed whole: shrink the bookkeeping to what was mapped)
    const size_t reserved = pool_n;
    for (size_t i = 0; i < got; ++i) {
      (void)hipMemUnmap((char*)pool.va + i * chunk, chunk);
      (void)hipMemRelease(pool.handles[i]);
    }
    (void)hipMemAddressFree(pool.va, reserved * chunk);
    (void)hipGetLastError();
    return nullptr;
  }
  std::vector<std::pair<float, size_t>> rate(got);
  for (size_t i = 0; i < got; ++i) {
    const float ms = probe_arena_ms((const float*)((char*)pool.va + i * chunk), chunk_tiles, nsens(ctx), 0, sink, ctx->stream);
    rate[i] = std::make_pair(ms > 0.0f ? ms : 1e30f, i);
  }
  std::sort(rate.begin(), rate.end());
  std::vector<char> keep(got, 0);
  for (size_t k = 0; k < need; ++k) keep[rate[k].second] = 1;
  // the pool's range goes (its reservation covers pool_n chunks, mapped or not); the kept chunks move into a range of their own
  {
    for (size_t i = 0; i < got; ++i) {
      (void)hipMemUnmap((char*)pool.va + i * chunk, chunk);
      if (!keep[i]) (void)hipMemRelease(pool.handles[i]);
    }
    (void)hipMemAddressFree(pool.va, pool_n * chunk);
    (void)hipGetLastError();
  }
  void* va = nullptr;
  kept.clear();
  bool ok = hipMemAddressReserve(&va, need * chunk, 0, nullptr, 0) == hipSuccess;
  size_t mapped = 0;
  for (size_t i = 0; ok && i < got; ++i) {
    if (!keep[i]) continue;
    ok = hipMemMap((char*)va + mapped * chunk, chunk, 0, pool.handles[i], 0) == hipSuccess;
    if (ok) {
      kept.push_back(pool.handles[i]);
      ++mapped;
    }
  }
  if (ok) ok = hipMemSetAccess(va, need * chunk, &acc, 1) == hipSuccess;
  float ms = -1.0f;
  if (ok) {
    ms = probe_arena_ms((const float*)va + head_floats, ntiles, nsens(ctx), g.tiles[0], sink, ctx->stream);
    ok = ms > 0.0f;
  }
  if (!ok) {
    for (size_t k = 0; k < mapped; ++k) (void)hipMemUnmap((char*)va + k * chunk, chunk);
    for (size_t i = 0; i < got; ++i)
      if (keep[i]) (void)hipMemRelease(pool.handles[i]);
    if (va) (void)hipMemAddressFree(va, need * chunk);
    (void)hipGetLastError();
    kept.clear();
    return nullptr;
  }
  *ms_out = ms;
  return (float*)va;
}

extern "C" {
static int ensure_tiled_lut(rgbdr_ctx* ctx)
{
  if (ctx->d_lut_tiled) return RGBDR_OK;
  const rgbdr_geometry& g = ctx->geo;
  const size_t ntiles = (size_t)g.tiles[0] * g.tiles[1] * (g.slab_tile_z1 - g.slab_tile_z0);
  const size_t layer = (size_t)g.tiles[0] * g.tiles[1] * nsens(ctx) * 3 * kTileVoxels;
  const size_t layers = (size_t)(g.slab_tile_z1 - g.slab_tile_z0) + 2 * (size_t)ctx->halo;
  const size_t bytes = layer * layers * sizeof(float);
  // Where the driver places this arena shifts the sweep time of integrate by up to 12 % (stable per allocation;
  // DESIGN.md 4.1): a zone of 13-19 GB of the device memory -- usually the one the first large allocation of a
  // process lands in -- streams at 5.9 TB/s, the rest at 6.6.  So the kernel's memory streams are timed on up to
  // RGBDR_ARENA_TRIALS candidate placements (1..16; default 16 for arenas of 1 GiB and more, else 1 = no probing)
  // and the fastest is kept.  Candidates are held while probing (otherwise the next allocation returns the same
  // place), so this transiently needs up to n x the arena; it stops at the first candidate at the fast level,
  // when less than arena + 4 GiB is free, or after 1 s + 0.12 s per GiB of arena.
  // (3 until round 4: on one box in four none of the first three candidates was fast, on one in six none of the first
  // eight; the loop stops at the first fast one, when memory runs short, or when its time is up, so the maximum only costs where it pays)
  int trials = bytes >= ((size_t)1 << 30) ? 16 : 1;
  if (const char* e = std::getenv("RGBDR_ARENA_TRIALS")) trials = std::atoi(e);
  if (trials > 16) trials = 16;
  if (trials < 1 || bytes < ((size_t)256 << 20)) trials = 1;  // small arenas: nothing to gain
  // four planes: window origins, the tiles' smallest and largest projected depths, footprint size class
  // (launch_tile_windows).  Allocated BEFORE any candidate is probed: a failure here must not leave a volume the
  // probe has scribbled over with tile states that still look valid.
  if (hipMalloc((void**)&ctx->d_win, 4 * ntiles * nsens(ctx) * sizeof(int32_t)) != hipSuccess) {
    (void)hipGetLastError();
    ctx->d_win = nullptr;
    return ctx->fail(RGBDR_ERR_HIP, "hipMalloc of the per-tile window words failed: out of device memory");
  }
  float* cand[16] = {nullptr};
  float* sink = ctx->d_tsdf_owned;  // the probe replays the TSDF store stream too: the volume is invalidated below
  float best_ms = 0.0f;
  int best = -1, got = 0;
  bool probed = false, reached_fast = false;
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int t = 0; t < trials; ++t) {
    size_t free_b = 0, total_b = 0;
    if (t > 0 && (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < bytes + ((size_t)4 << 30))) break;
    // (plain hipMalloc on purpose: physically contiguous arenas -- hipExtMallocWithFlags + hipDeviceMallocContiguous --
    // stream at two levels instead of three, but allocating and freeing them next to live buffers made parity tests
    // fail intermittently and once hung a process for 15 minutes: profiles/r03_notes)
    if (hipMalloc((void**)&cand[t], bytes) != hipSuccess) {
      (void)hipGetLastError();
      cand[t] = nullptr;
      if (t == 0) {
        (void)hipFree(ctx->d_win);
        ctx->d_win = nullptr;
        return ctx->fail(RGBDR_ERR_HIP, "hipMalloc of the inverse-LUT arena failed: out of device memory");
      }
      break;
    }
    got = t + 1;
    if (trials == 1) {
      best = 0;
      break;
    }
    const float ms = probe_arena_ms(cand[t] + layer * ctx->halo, ntiles, nsens(ctx), g.tiles[0], sink, ctx->stream);
    probed = true;
    ctx->arena_probe_ms[t] = ms;
    if (ms > 0.0f && (best < 0 || ms < best_ms)) {
      best = t;
      best_ms = ms;
    }
    // stop at the first candidate that streams at the fastest level seen on this hardware
    // (>= 6.6 TB/s for LUT reads + TSDF stores; the others are 5.9-6.5 TB/s)
    const double stream_bytes = (double)ntiles * ((double)nsens(ctx) * 3 + 1) * kTileVoxels * sizeof(float);
    if (ms > 0.0f && stream_bytes / (ms * 1e-3) >= 6.6e12) {
      reached_fast = true;
      break;
    }
    // time budget: 1 s + 0.12 s per GiB of arena (a candidate costs its hipMalloc, which grows with its size: on one box
    // three candidates of the 8-sensor arena -- 25.8 GB -- used up a flat 1 s, all three at the slow level)
    clock_gettime(CLOCK_MONOTONIC, &t1);
    const double budget_s = 1.0 + 0.12 * (double)(bytes >> 30);
    if ((double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec) > budget_s) break;
  }
  if (best < 0) best = 0;
  ctx->arena_trials = got;
  ctx->arena_chosen = best;
  int freed = 0;
  for (int t = 0; t < got; ++t)
    if (t != best) {
      (void)hipFree(cand[t]);
      ++freed;
    }
  // No plain candidate at the fast level (or only one could be held): an arena of the fastest physical chunks, kept if its
  // replay beats the best plain one's (8 sensors into 1024^3, no room to shop: replay 16.52 against 16.61 ms, the sweep
  // itself 16.91 against 17.47 -- a plain arena that is slow in parts costs the kernel more than it costs the replay).  RGBDR_ARENA_CHUNKS=0 never, =force always (tests); RGBDR_ARENA_CHUNK_MB
  // sets the chunk size (default 1024).
  float* base = cand[best];
  ctx->arena_chunks = 0;
  ctx->arena_chunk_ms = 0.0f;
  {
    const char* ce = std::getenv("RGBDR_ARENA_CHUNKS");
    const bool force = ce && std::strcmp(ce, "force") == 0, never = ce && std::strcmp(ce, "0") == 0;
    size_t chunk = (size_t)1 << 30;
    if (const char* cm = std::getenv("RGBDR_ARENA_CHUNK_MB")) chunk = (size_t)(std::atoi(cm) > 0 ? std::atoi(cm) : 1024) << 20;
    if (!never && (force || (trials > 1 && probed && !reached_fast)) && bytes >= chunk) {
      float ms_c = 0.0f;
      std::vector<hipMemGenericAllocationHandle_t> kept;
      float* va = build_chunk_arena(ctx, bytes, layer * (size_t)ctx->halo, ntiles, sink, chunk, &ms_c, kept);
      if (va && (force || !(best_ms > 0.0f) || ms_c < best_ms)) {
        (void)hipFree(cand[best]);
        ++freed;
        base = va;
        best_ms = ms_c;
        ctx->lut_vmm_va = va;
        ctx->lut_vmm_chunk = chunk;
        ctx->lut_vmm_handles = kept;
        ctx->arena_chunks = (int)kept.size();
        ctx->arena_chunk_ms = ms_c;
        probed = true;
      } else if (va) {
        for (size_t k = 0; k < kept.size(); ++k) {
          (void)hipMemUnmap((char*)va + k * chunk, chunk);
          (void)hipMemRelease(kept[k]);
        }
        (void)hipMemAddressFree(va, kept.size() * chunk);
        (void)hipGetLastError();
        ++freed;
      }
    }
  }
  // Releasing that much memory slows the device down for a moment (the driver wipes released VRAM
  // in the background): wait, at most 2 s, until the kept arena streams as it did when it was chosen.
  if (freed > 0 && best_ms > 0.0f) {
    for (int k = 0; k < 40; ++k) {
      const float ms = probe_arena_ms(base + layer * ctx->halo, ntiles, nsens(ctx), g.tiles[0], sink, ctx->stream);
      if (!(ms > best_ms * 1.01f)) break;
      struct timespec ts = {0, 50000000};
      nanosleep(&ts, nullptr);
    }
  }
  ctx->d_lut_tiled_base = base;
  ctx->d_lut_tiled = ctx->d_lut_tiled_base + layer * ctx->halo;
  if (probed) {  // the replay stored into the volume: clear it again, forget recorded clears, nothing is integrated
    HIPCHK(hipMemsetAsync(ctx->d_tsdf_owned, 0, ntiles * kTileVoxels * sizeof(float), ctx->stream));
    { int rc_ = bump_clear_epoch(ctx); if (rc_ != RGBDR_OK) return rc_; }
    ctx->integrated = false;
  }
  HIPCHK(hipMemsetAsync(ctx->d_win, 0, 4 * ntiles * nsens(ctx) * sizeof(int32_t), ctx->stream));
  return RGBDR_OK;
}

int rgbdr_set_inverse_calibration(rgbdr_ctx* ctx, int sensor, const rgbdr_lut* inv)
try {
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (sensor < 0 || sensor >= nsens(ctx)) return ctx->fail(RGBDR_ERR_OUT_OF_RANGE, "sensor index out of range");
  if (!inv || !inv->data) return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, "null inverse calibration volume");
  if (const char* e = lut_res_error(inv->res)) return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, e);
  HIPCHK(hipSetDevice(ctx->device));
  const rgbdr_geometry& g = ctx->geo;
  const int X = inv->res[0], Y = inv->res[1], Z = inv->res[2];
  const float4* host = (const float4*)inv->data;
  (void)hipFree(ctx->d_lut_generic[sensor]);
  ctx->d_lut_generic[sensor] = nullptr;
  ctx->inv_set[sensor] = false;
  ctx->inv_resampled[sensor] = false;
  for (int a = 0; a < 3; ++a) ctx->inv_res[sensor][a] = inv->res[a];
  // Every sensor of a context is resident in the same layout, so LUTs of any mix of resolutions work
  // together: the grid layout (a 1:1 LUT re-tiled, any other resolution resampled at the voxel centres
  // once -- the lookup of tsdf_integration.vs:31 is static between frames) unless RGBDR_FLAG_NO_RESAMPLE
  // asks for the file layout, or the arena did not fit when the first sensor was set.
  bool others_tiled = false, others_file = false;
  for (int i = 0; i < nsens(ctx); ++i) {
    if (i == sensor || !ctx->inv_set[i]) continue;
    others_tiled = others_tiled || ctx->inv_tiled[i];
    others_file = others_file || !ctx->inv_tiled[i];
  }
  bool grid_layout = !(ctx->cfg.flags & RGBDR_FLAG_NO_RESAMPLE) && !others_file;
  if (grid_layout && ensure_tiled_lut(ctx) != RGBDR_OK) {
    (void)hipGetLastError();
    if (others_tiled) return RGBDR_ERR_HIP;  // message set by ensure_tiled_lut; cannot happen: the arena exists already
    grid_layout = false;                     // the arena does not fit: keep the file's volume, sample per frame
  }
  const size_t row = (size_t)X * Y;
  if (grid_layout && lut_is_one_to_one(inv->res, g.res_volume)) {
    // stage whole tile layers through a bounded scratch buffer
    const int chunk_layers = 8;
    DevScratch tmp;
    HIPCHK(hipMalloc(&tmp.p, row * kTile * chunk_layers * sizeof(float4)));
    const LutExtent ext = lut_extent(ctx);
    for (int tz = ext.t0; tz < ext.t1; tz += chunk_layers) {
      const int tz_end = tz + chunk_layers < ext.t1 ? tz + chunk_layers : ext.t1;
      const int vz0 = tz * kTile;
      int vz1 = tz_end * kTile;
      if (vz1 > Z) vz1 = Z;
      HIPCHK(hipMemcpyAsync(tmp.p, host + row * vz0, row * (size_t)(vz1 - vz0) * sizeof(float4), hipMemcpyHostToDevice,
                            ctx->stream));
      float* dst = ext.dst + (size_t)(tz - ext.t0) * g.tiles[0] * g.tiles[1] * nsens(ctx) * 3 * kTileVoxels;
      launch_tile_lut(tmp.as<float4>(), X, Y, Z, vz0, g.tiles[0], g.tiles[1], tz, tz_end - tz, sensor, nsens(ctx), dst,
                      ctx->stream);
      LAUNCHCHK("tile_lut");
      HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    launch_tile_windows(ctx->d_lut_tiled, ctx->cfg.depth_w, ctx->cfg.depth_h,
                        g.tiles[0] * g.tiles[1] * (g.slab_tile_z1 - g.slab_tile_z0), sensor, nsens(ctx), ctx->d_win,
                        ctx->stream);
    LAUNCHCHK("tile_windows");
    ctx->bgmax_for = -1;  // the skip verdicts of the current frame were taken from the old planes
    ctx->inv_tiled[sensor] = true;
  } else {
    int lo, hi;
    const LutExtent ext = lut_extent(ctx);
    lut_z_range(Z, g.res_volume[2], ext.vz0, ext.vz1, &lo, &hi);
    const size_t cnt = row * (size_t)(hi - lo + 1);
    HIPCHK(hipMalloc((void**)&ctx->d_lut_generic[sensor], cnt * sizeof(float4)));
    HIPCHK(hipMemcpyAsync(ctx->d_lut_generic[sensor], host + row * lo, cnt * sizeof(float4), hipMemcpyHostToDevice,
                          ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->zoff[sensor] = lo;
    ctx->inv_tiled[sensor] = false;
    if (grid_layout) {
      launch_resample_lut(ctx->d_lut_generic[sensor], X, Y, Z, lo, g.res_volume[0], g.res_volume[1], g.res_volume[2],
                          g.tiles[0], g.tiles[1], ext.t0, ext.t1 - ext.t0, sensor, nsens(ctx), ext.dst, ctx->stream);
      LAUNCHCHK("resample_lut");
      launch_tile_windows(ctx->d_lut_tiled, ctx->cfg.depth_w, ctx->cfg.depth_h,
                          g.tiles[0] * g.tiles[1] * (g.slab_tile_z1 - g.slab_tile_z0), sensor, nsens(ctx), ctx->d_win,
                          ctx->stream);
      LAUNCHCHK("tile_windows");
      ctx->bgmax_for = -1;  // the skip verdicts of the current frame were taken from the old planes
      HIPCHK(hipStreamSynchronize(ctx->stream));
      (void)hipFree(ctx->d_lut_generic[sensor]);
      ctx->d_lut_generic[sensor] = nullptr;
      ctx->inv_tiled[sensor] = true;
      ctx->inv_resampled[sensor] = true;
    }
  }
  ctx->inv_set[sensor] = true;
  return RGBDR_OK;
}
RGBDR_CONTAIN(ctx)

int rgbdr_device_calibration(rgbdr_ctx* ctx, int sensor, rgbdr_calibration_device_view* out)
try {
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (!out) return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, "null view");
  if (sensor < 0 || sensor >= nsens(ctx)) return ctx->fail(RGBDR_ERR_OUT_OF_RANGE, "sensor index out of range");
  if (!ctx->have_calib[sensor]) return ctx->fail(RGBDR_ERR_STATE, "rgbdr_device_calibration before set_calibration of this sensor");
  out->cv_xyz = ctx->d_cv_xyz[sensor];
  out->cv_uv = ctx->d_cv_uv[sensor];
  for (int a = 0; a < 3; ++a) {
    out->xyz_res[a] = ctx->xyz_res[sensor][a];
    out->uv_res[a] = ctx->uv_res[sensor][a];
    out->inv_res[a] = ctx->inv_set[sensor] ? ctx->inv_res[sensor][a] : 0u;
  }
  out->depth_limits[0] = ctx->min_ds[sensor];
  out->depth_limits[1] = ctx->max_ds[sensor];
  out->stream = (void*)ctx->stream;
  return RGBDR_OK;
}
RGBDR_CONTAIN(ctx)

static int read_lut_file(rgbdr_ctx* ctx, const char* path, size_t rec_bytes, rgbdr_lut* lut, std::vector<char>* buf)
{
  FILE* f = std::fopen(path, "rb");
  if (!f) return ctx->fail(RGBDR_ERR_IO, std::string("cannot open ") + path);
  bool ok = std::fread(lut->res, 4, 3, f) == 3 && std::fread(lut->depth_limits, 4, 2, f) == 2;
  if (ok) {
    // the header is checked against what the file holds BEFORE anything is sized by it: a corrupt or truncated file is
    // an I/O error, not an attempt to allocate res[0] * res[1] * res[2] records (the reference reads header and payload
    // with its fread results ignored: calibration_volume.hpp:60-78)
    long here = std::ftell(f), end = -1;
    if (here >= 0 && std::fseek(f, 0, SEEK_END) == 0) end = std::ftell(f);
    ok = here >= 0 && end >= here && std::fseek(f, here, SEEK_SET) == 0;
    const unsigned __int128 want = (unsigned __int128)lut->res[0] * lut->res[1] * lut->res[2] * rec_bytes;
    if (ok && (lut->res[0] == 0 || lut->res[1] == 0 || lut->res[2] == 0 || want > (unsigned __int128)(end - here))) {
      std::fclose(f);
      return ctx->fail(RGBDR_ERR_IO, std::string(path) + ": header says " + std::to_string(lut->res[0]) + " x " + std::to_string(lut->res[1]) +
                                         " x " + std::to_string(lut->res[2]) + " records of " + std::to_string(rec_bytes) + " bytes, the file holds " +
                                         std::to_string(end - here) + " bytes after the header");
    }
    if (ok) {
      const size_t n = (size_t)want;
      buf->resize(n);
      ok = std::fread(buf->data(), 1, n, f) == n;
    }
  }
  std::fclose(f);
  if (!ok) return ctx->fail(RGBDR_ERR_IO, std::string("short read from ") + path);
  lut->data = buf->data();
  return RGBDR_OK;
}

int rgbdr_load_calibration_files(rgbdr_ctx* ctx, int sensor, const char* pxyz, const char* puv, const char* pinv)
try {
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if ((pxyz == nullptr) != (puv == nullptr))
    return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, "cv_xyz and cv_uv must be given together");
  if (pxyz) {
    rgbdr_lut a{}, b{};
    std::vector<char> ba, bb;
    int rc = read_lut_file(ctx, pxyz, 12, &a, &ba);
    if (rc != RGBDR_OK) return rc;
    rc = read_lut_file(ctx, puv, 8, &b, &bb);
    if (rc != RGBDR_OK) return rc;
    rc = rgbdr_set_calibration(ctx, sensor, &a, &b);
    if (rc != RGBDR_OK) return rc;
  }
  if (pinv) {
    rgbdr_lut c{};
    std::vector<char> bc;
    int rc = read_lut_file(ctx, pinv, 16, &c, &bc);
    if (rc != RGBDR_OK) return rc;
    rc = rgbdr_set_inverse_calibration(ctx, sensor, &c);
    if (rc != RGBDR_OK) return rc;
  }
  return RGBDR_OK;
}
RGBDR_CONTAIN(ctx)

int rgbdr_synth_inverse_calibration(rgbdr_ctx* ctx, int sensor, const rgbdr_pinhole* cam)
try {
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (sensor < 0 || sensor >= nsens(ctx)) return ctx->fail(RGBDR_ERR_OUT_OF_RANGE, "sensor index out of range");
  if (!cam) return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, "null pinhole");
  HIPCHK(hipSetDevice(ctx->device));
  const rgbdr_geometry& g = ctx->geo;
  const uint32_t r[3] = {(uint32_t)g.res_volume[0], (uint32_t)g.res_volume[1], (uint32_t)g.res_volume[2]};
  if (!lut_is_one_to_one(r, g.res_volume))
    return ctx->fail(RGBDR_ERR_STATE, "synthetic inverse LUT needs a grid whose voxel centres hit texel centres exactly");
  if (ctx->cfg.flags & RGBDR_FLAG_NO_RESAMPLE)
    return ctx->fail(RGBDR_ERR_STATE, "the synthetic (benchmark) inverse LUT is written in the grid layout only: not with RGBDR_FLAG_NO_RESAMPLE");
  for (int i = 0; i < nsens(ctx); ++i)
    if (i != sensor && ctx->inv_set[i] && !ctx->inv_tiled[i])
      return ctx->fail(RGBDR_ERR_STATE, "other sensors hold file-layout inverse LUTs (the grid-layout arena did not fit when they were set)");
  int rc = ensure_tiled_lut(ctx);
  if (rc != RGBDR_OK) return rc;
  const LutExtent ext = lut_extent(ctx);
  launch_synth_inverse(*cam, ctx->cfg.depth_w, ctx->cfg.depth_h, ctx->cfg.bbox_min, ctx->cfg.bbox_max, g.res_volume[0],
                       g.res_volume[1], g.res_volume[2], g.tiles[0], g.tiles[1], ext.t0, ext.t1 - ext.t0, sensor,
                       nsens(ctx), ext.dst, ctx->stream);
  LAUNCHCHK("synth_inverse");
  launch_tile_windows(ctx->d_lut_tiled, ctx->cfg.depth_w, ctx->cfg.depth_h,
                      g.tiles[0] * g.tiles[1] * (g.slab_tile_z1 - g.slab_tile_z0), sensor, nsens(ctx), ctx->d_win,
                      ctx->stream);
  LAUNCHCHK("tile_windows");
  ctx->bgmax_for = -1;  // the skip verdicts of the current frame were taken from the old planes
  for (int a = 0; a < 3; ++a) ctx->inv_res[sensor][a] = r[a];
  (void)hipFree(ctx->d_lut_generic[sensor]);
  ctx->d_lut_generic[sensor] = nullptr;
  ctx->inv_tiled[sensor] = true;
  ctx->inv_set[sensor] = true;
  return RGBDR_OK;
}
RGBDR_CONTAIN(ctx)

// CalibrationInverter::calculateInverseVolumes on the device (kernels_invert.hip)
static void fill_invert_params(rgbdr_ctx* ctx, int sensor, const int32_t vol_res[3], int window, InvertParams* p)
{
  *p = InvertParams{};
  p->xyz = ctx->d_cv_xyz[sensor];
  p->rx = (int)ctx->xyz_res[sensor][0];
  p->ry = (int)ctx->xyz_res[sensor][1];
  p->rz = (int)ctx->xyz_res[sensor][2];
  std::memcpy(p->planes, ctx->planes[sensor], sizeof(p->planes));
  for (int a = 0; a < 3; ++a) {
    const float vstep = 1.0f / (float)vol_res[a];
    p->step[a] = (ctx->cfg.bbox_max[a] - ctx->cfg.bbox_min[a]) * vstep;
    p->start[a] = ctx->cfg.bbox_min[a] + p->step[a] * 0.5f;
  }
  p->X = vol_res[0];
  p->Y = vol_res[1];
  p->TX = (vol_res[0] + kTile - 1) / kTile;
  p->TY = (vol_res[1] + kTile - 1) / kTile;
  // R = 2 certifies 88 % of the voxels at once and leaves 12 % to k_invert_retry (68 ms per sensor at 512^3); R = 3 certifies
  // all but 1e-5 but scans 343 instead of 125 samples for everyone (87 ms)
  p->window = window < 1 ? 2 : window;
  p->folded = ctx->lattice_folded[sensor] ? 1 : 0;
  p->sensor = sensor;
  p->N = ctx->cfg.num_sensors;
}

// One certified search over the z rows [p.z0, p.z0 + p.nz): the local search and the exhaustive scan of whatever it could
// not certify (kernels_invert.hip).  Rows are taken in pieces of at most 64 so that the list of uncertified voxels (one
// word per voxel of a piece: it can never overflow) stays small.
// An API call that searches in several chunks (rgbdr_generate_inverse_lut streams 64 rows at a time to the host) hands in
// one scratch allocation for all of them (`shared`, sized by the first = largest chunk) and the counters of
// rgbdr_inverse_search_stats add up over the chunks: the entry point zeroes them, not this function.
static int run_invert(rgbdr_ctx* ctx, InvertParams p, int sensor, DevScratch* shared = nullptr)
{
  const int z_begin = p.z0, z_end = p.z0 + p.nz;
  const int piece = 64;
  const size_t row = (size_t)p.X * p.Y;
  const size_t head = 6;  // [2] the two lists' lengths, [4] two 64-bit counters
  const size_t per_piece = row * (size_t)std::min(piece, p.nz);
  const size_t words = head + 3 * per_piece;  // exhaustive list: a word per voxel; retry list: two
  DevScratch own;
  DevScratch& aux = shared ? *shared : own;
  if (!aux.p) HIPCHK(hipMalloc(&aux.p, words * sizeof(unsigned)));
  unsigned* base = aux.as<unsigned>();
  unsigned* count = base;
  unsigned long long* stats = (unsigned long long*)(base + 2);  // 8-byte aligned
  unsigned* todo = base + head;
  HIPCHK(hipMemsetAsync(base, 0, head * sizeof(unsigned), ctx->stream));
  p.todo = todo;
  p.todo_count = count;
  p.retry = todo + per_piece;
  p.retry_count = count + 1;
  p.stats = stats;
  float* tiled = p.out_tiled;
  float4* linear = p.out_linear;
  for (int z = z_begin; z < z_end; z += piece) {
    p.z0 = z;
    p.nz = std::min(piece, z_end - z);
    const size_t done_tiles = (size_t)((z - z_begin) / kTile) * p.TX * p.TY;
    p.out_tiled = tiled ? tiled + done_tiles * (size_t)p.N * 3 * kTileVoxels : nullptr;
    p.out_linear = linear ? linear + row * (size_t)(z - z_begin) : nullptr;
    HIPCHK(hipMemsetAsync(count, 0, 2 * sizeof(unsigned), ctx->stream));
    launch_invert_lut(p, ctx->stream);
    LAUNCHCHK("invert_lut");
    launch_invert_retry(p, ctx->stream);
    LAUNCHCHK("invert_retry");
    launch_invert_exhaustive(p, ctx->stream);
    LAUNCHCHK("invert_exhaustive");
  }
  unsigned long long h[2] = {0, 0};
  HIPCHK(hipMemcpyAsync(h, stats, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ctx->inv_search_widened[sensor] += h[0];
  ctx->inv_search_exhaustive[sensor] += h[1];
  return RGBDR_OK;
}

int rgbdr_inverse_search_stats(rgbdr_ctx* ctx, int sensor, uint64_t* widened, uint64_t* exhaustive)
try {
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (sensor < 0 || sensor >= nsens(ctx)) return ctx->fail(RGBDR_ERR_OUT_OF_RANGE, "sensor index out of range");
  if (widened) *widened = ctx->inv_search_widened[sensor];
  if (exhaustive) *exhaustive = ctx->inv_search_exhaustive[sensor];
  return RGBDR_OK;
}
RGBDR_CONTAIN(ctx)

int rgbdr_compute_inverse_calibration(rgbdr_ctx* ctx, int sensor, int window)
try {
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (sensor < 0 || sensor >= nsens(ctx)) return ctx->fail(RGBDR_ERR_OUT_OF_RANGE, "sensor index out of range");
  if (!ctx->have_calib[sensor]) return ctx->fail(RGBDR_ERR_STATE, "compute_inverse_calibration before set_calibration");
  if (window > 8) return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, "window radius must be <= 8");
  HIPCHK(hipSetDevice(ctx->device));
  // every sensor of a context is resident in one layout (set_inverse_calibration): with RGBDR_FLAG_NO_RESAMPLE,
  // or next to sensors that hold the file layout, the generated volume is kept as an x-fastest RGBA32F volume at
  // the grid resolution (the rows this slab samples) and looked up per frame like an uploaded one
  bool file_layout = (ctx->cfg.flags & RGBDR_FLAG_NO_RESAMPLE) != 0;
  for (int i = 0; i < nsens(ctx); ++i)
    if (i != sensor && ctx->inv_set[i] && !ctx->inv_tiled[i]) file_layout = true;
  const rgbdr_geometry& g = ctx->geo;
  InvertParams p;
  fill_invert_params(ctx, sensor, g.res_volume, window, &p);
  if (file_layout) {
    for (int i = 0; i < nsens(ctx); ++i)
      if (i != sensor && ctx->inv_set[i] && ctx->inv_tiled[i])
        return ctx->fail(RGBDR_ERR_STATE, "other sensors hold grid-layout inverse LUTs generated on the device: set them again");
    int lo, hi;
    const LutExtent fext = lut_extent(ctx);
    lut_z_range(g.res_volume[2], g.res_volume[2], fext.vz0, fext.vz1, &lo, &hi);
    lo = (lo / kTile) * kTile;  // whole tiles from the volume's tile grid: the search is seeded per tile
    const size_t frow = (size_t)g.res_volume[0] * g.res_volume[1];
    (void)hipFree(ctx->d_lut_generic[sensor]);
    ctx->d_lut_generic[sensor] = nullptr;
    ctx->inv_set[sensor] = false;
    HIPCHK(hipMalloc((void**)&ctx->d_lut_generic[sensor], frow * (size_t)(hi - lo + 1) * sizeof(float4)));
    p.z0 = lo;
    p.nz = hi - lo + 1;
    p.out_linear = ctx->d_lut_generic[sensor];
    p.out_tiled = nullptr;
    ctx->inv_search_widened[sensor] = ctx->inv_search_exhaustive[sensor] = 0;
    { int rc_ = run_invert(ctx, p, sensor); if (rc_ != RGBDR_OK) return rc_; }
    for (int a = 0; a < 3; ++a) ctx->inv_res[sensor][a] = (uint32_t)g.res_volume[a];
    ctx->zoff[sensor] = lo;
    ctx->inv_tiled[sensor] = false;
    ctx->inv_resampled[sensor] = false;
    ctx->inv_set[sensor] = true;
    return RGBDR_OK;
  }
  int rc = ensure_tiled_lut(ctx);
  if (rc != RGBDR_OK) return rc;
  const LutExtent ext = lut_extent(ctx);
  p.z0 = ext.vz0;
  p.nz = ext.vz1 - ext.vz0;
  p.out_tiled = ext.dst;
  ctx->inv_search_widened[sensor] = ctx->inv_search_exhaustive[sensor] = 0;
  { int rc_ = run_invert(ctx, p, sensor); if (rc_ != RGBDR_OK) return rc_; }
  launch_tile_windows(ctx->d_lut_tiled, ctx->cfg.depth_w, ctx->cfg.depth_h,
                      g.tiles[0] * g.tiles[1] * (g.slab_tile_z1 - g.slab_tile_z0), sensor, nsens(ctx), ctx->d_win,
                      ctx->stream);
  LAUNCHCHK("tile_windows");
  ctx->bgmax_for = -1;  // the skip verdicts of the current frame were taken from the old planes
  HIPCHK(hipStreamSynchronize(ctx->stream));
  (void)hipFree(ctx->d_lut_generic[sensor]);
  ctx->d_lut_generic[sensor] = nullptr;
  for (int a = 0; a < 3; ++a) ctx->inv_res[sensor][a] = (uint32_t)g.res_volume[a];
  ctx->inv_tiled[sensor] = true;
  ctx->inv_resampled[sensor] = false;
  ctx->inv_set[sensor] = true;
  return RGBDR_OK;
}
RGBDR_CONTAIN(ctx)

int rgbdr_generate_inverse_lut(rgbdr_ctx* ctx, int sensor, const uint32_t res[3], int window, float* dst)
try {
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (sensor < 0 || sensor >= nsens(ctx)) return ctx->fail(RGBDR_ERR_OUT_OF_RANGE, "sensor index out of range");
  if (!res || !dst) return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, "null resolution / destination");
  if (const char* e = lut_res_error(res)) return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, e);
  if (!ctx->have_calib[sensor]) return ctx->fail(RGBDR_ERR_STATE, "generate_inverse_lut before set_calibration");
  if (window > 8) return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, "window radius must be <= 8");
  HIPCHK(hipSetDevice(ctx->device));
  const int32_t vr[3] = {(int32_t)res[0], (int32_t)res[1], (int32_t)res[2]};
  InvertParams p;
  fill_invert_params(ctx, sensor, vr, window, &p);
  const size_t row = (size_t)res[0] * res[1];
  const int chunk = 64;
  DevScratch tmp, aux;
  HIPCHK(hipMalloc(&tmp.p, row * chunk * sizeof(float4)));
  ctx->inv_search_widened[sensor] = ctx->inv_search_exhaustive[sensor] = 0;  // "the sensor's last search": this whole call
  for (int z = 0; z < (int)res[2]; z += chunk) {
    p.z0 = z;
    p.nz = z + chunk <= (int)res[2] ? chunk : (int)res[2] - z;
    p.out_linear = tmp.as<float4>();
    { int rc_ = run_invert(ctx, p, sensor, &aux); if (rc_ != RGBDR_OK) return rc_; }
    HIPCHK(hipMemcpyAsync(dst + row * 4 * (size_t)z, tmp.p, row * (size_t)p.nz * sizeof(float4), hipMemcpyDeviceToHost,
                          ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
  }
  return RGBDR_OK;
}
RGBDR_CONTAIN(ctx)

int rgbdr_readback_inverse_calibration(rgbdr_ctx* ctx, int sensor, int z0, int z1, float* dst)
try {
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (!dst) return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, "null destination");
  if (sensor < 0 || sensor >= nsens(ctx)) return ctx->fail(RGBDR_ERR_OUT_OF_RANGE, "sensor index out of range");
  if (!ctx->inv_set[sensor]) return ctx->fail(RGBDR_ERR_STATE, "inverse calibration of this sensor is not set");
  HIPCHK(hipSetDevice(ctx->device));
  const rgbdr_geometry& g = ctx->geo;
  const bool tiled = ctx->inv_tiled[sensor];
  const int X = tiled ? g.res_volume[0] : (int)ctx->inv_res[sensor][0];
  const int Y = tiled ? g.res_volume[1] : (int)ctx->inv_res[sensor][1];
  const size_t row = (size_t)X * Y;
  if (tiled) {
    if (z0 < g.slab_voxel_z0 || z1 > g.slab_voxel_z1 || z0 >= z1)
      return ctx->fail(RGBDR_ERR_OUT_OF_RANGE, "z rows outside this context's slab");
    DevScratch tmp;
    HIPCHK(hipMalloc(&tmp.p, row * (size_t)(z1 - z0) * sizeof(float4)));
    launch_untile_lut(ctx->d_lut_tiled, X, Y, g.tiles[0], g.tiles[1], g.slab_tile_z0, z0, z1, sensor, nsens(ctx),
                      tmp.as<float4>(), ctx->stream);
    LAUNCHCHK("untile_lut");
    HIPCHK(hipMemcpyAsync(dst, tmp.p, row * (size_t)(z1 - z0) * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
  } else {
    int lo, hi;
    const LutExtent ext = lut_extent(ctx);
    lut_z_range((int)ctx->inv_res[sensor][2], g.res_volume[2], ext.vz0, ext.vz1, &lo, &hi);
    if (z0 < lo || z1 > hi + 1 || z0 >= z1) return ctx->fail(RGBDR_ERR_OUT_OF_RANGE, "texel rows not resident");
    HIPCHK(hipMemcpyAsync(dst, ctx->d_lut_generic[sensor] + row * (size_t)(z0 - lo),
                          row * (size_t)(z1 - z0) * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
  }
  return RGBDR_OK;
}
RGBDR_CONTAIN(ctx)

int rgbdr_settle(rgbdr_ctx* ctx, float max_seconds, float* stream_ms)
try {
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (!ctx->d_lut_tiled) return ctx->fail(RGBDR_ERR_STATE, "settle before the inverse LUTs were set");
  // the budget bounds the loop below: a NaN would never compare as exceeded (a hang found by tests/test_chaos_gpu.py on a
  // volume too small to ever stream at the "steady" rate), and no caller means to wait longer than a minute
  if (!(max_seconds >= 0.0f)) return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, "settle: the budget must be a number of seconds >= 0");
  const double budget = max_seconds < 60.0f ? (double)max_seconds : 60.0;
  HIPCHK(hipSetDevice(ctx->device));
  { int rc_ = sync_all(ctx); if (rc_ != RGBDR_OK) return rc_; }
  const rgbdr_geometry& g = ctx->geo;
  const size_t ntiles = (size_t)g.tiles[0] * g.tiles[1] * (g.slab_tile_z1 - g.slab_tile_z0);
  { int rc_ = bump_clear_epoch(ctx); if (rc_ != RGBDR_OK) return rc_; }  // the replay stores into the volume ...
  ctx->integrated = false;  // ... whose contents are undefined until the next integrate
  // "steady" = the replay streams at the fastest level this hardware shows (>= 6.55 TB/s), or,
  // for an arena at one of the slower placements, the budget is used up.  (Agreement between
  // consecutive replays is not enough: a long wipe slows them all alike.)
  const double stream_bytes = (double)ntiles * ((double)nsens(ctx) * 3 + 1) * kTileVoxels * sizeof(float);
  float cur = -1.0f;
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (;;) {
    cur = probe_arena_ms(ctx->d_lut_tiled, ntiles, nsens(ctx), g.tiles[0], ctx->d_tsdf_owned, ctx->stream);
    if (cur < 0.0f) return ctx->fail(RGBDR_ERR_HIP, "settle: the stream replay failed");
    if (stream_bytes / (cur * 1e-3) >= 6.55e12) break;
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (!((double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec) <= budget)) break;
    struct timespec ts = {0, 50000000};
    nanosleep(&ts, nullptr);
  }
  if (stream_ms) *stream_ms = cur;
  return RGBDR_OK;
}
RGBDR_CONTAIN(ctx)

int rgbdr_get_arena_chunks(const rgbdr_ctx* ctx, int* chunks, float* ms)
try {
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (chunks) *chunks = ctx->arena_chunks;
  if (ms) *ms = ctx->arena_chunk_ms;
  return RGBDR_OK;
}
RGBDR_CONTAIN(nullptr)

int rgbdr_get_arena_probe(const rgbdr_ctx* ctx, float ms[16], int* trials, int* chosen)
try {
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (ms) std::memcpy(ms, ctx->arena_probe_ms, sizeof(ctx->arena_probe_ms));
  if (trials) *trials = ctx->arena_trials;
  if (chosen) *chosen = ctx->arena_chosen;
  return RGBDR_OK;
}
RGBDR_CONTAIN(nullptr)

}  // extern "C"
