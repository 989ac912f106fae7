// api.cpp -- the C ABI of include/rgbdr.h: context, device memory, call order.
// Host-side mirror of NetKinectArray / CalibVolumes / ReconIntegration state;
// every GL texture unit / SSBO binding of the reference is a
// device pointer owned by the context here.  There is no CPU fallback: without a
// HIP device rgbdr_create fails with RGBDR_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "context.hpp"

using namespace rgbdr;

namespace rgbdr {
thread_local std::string g_create_error;

// Whatever writes the volume without keeping tile_state (full sweep, generic-LUT sweep, stream
// replay) or changes -limit invalidates every recorded "this tile already holds -limit".
int bump_clear_epoch(rgbdr_ctx* ctx)
{
  if (++ctx->clear_epoch == 0) {  // wrapped: forget every recorded clear
    const rgbdr_geometry& g = ctx->geo;
    const size_t n = (size_t)g.tiles[0] * g.tiles[1] * (size_t)(g.slab_tile_z1 - g.slab_tile_z0);
    if (ctx->d_tile_state) HIPCHK(hipMemsetAsync(ctx->d_tile_state, 0, n * sizeof(uint32_t), ctx->stream));
    ctx->clear_epoch = 1;
  }
  return RGBDR_OK;
}

// bytes of one sensor's colour frame as the caller hands it over (NetKinectArray.cpp:120-131)
static size_t color_frame_bytes(const rgbdr_config& c)
{
  const size_t blocks = (size_t)((c.color_w + 3) / 4) * ((c.color_h + 3) / 4);
  if (c.compress_rgb == 1) return blocks * 8;
  if (c.compress_rgb == 5) return blocks * 16;
  return (size_t)c.color_w * c.color_h * 3;
}
static void free_volume(rgbdr_ctx* c)
{
  (void)hipFree(c->d_tsdf_base);
  (void)hipFree(c->d_linear);
  if (c->fill_stream) (void)hipStreamSynchronize(c->fill_stream);
  c->fill_side = false;
  c->ev_fill_rec[0] = c->ev_fill_rec[1] = false;
  (void)hipFree(c->d_view_base);
  c->d_view_base = nullptr;
  c->vbuf = 0;
  (void)hipFree(c->d_peels);
  (void)hipFree(c->d_peel_near);
  c->d_peels = nullptr;
  c->d_peel_near = nullptr;
  c->peel_pixels = c->peel_near_cap = 0;
  (void)hipFree(c->d_fill);
  (void)hipFree(c->d_fill_tabs);
  c->d_fill_tabs = nullptr;
  c->fill_tab_w = c->fill_tab_h = 0;
  c->d_view = c->d_fill = nullptr;
  c->view_pixels = c->fill_floats = 0;
  c->view_w = c->view_h = c->filled_w = c->filled_h = 0;
  c->integrated = false;
  for (int b = 0; b < 2; ++b)
    for (int f = 0; f < 2; ++f) {
      (void)hipFree(c->d_stage[b][f]);
      c->d_stage[b][f] = nullptr;
    }
  rgbdr::destroy_peer_state(c);  // the staging sets the neighbours mapped are gone: export and set the peers again
  c->stage_target = -1;
  c->halo_begun = c->halo_staged = false;  // a resize between begin_step and exchange_async: the exchange has nothing to send
  c->halo_done_rec[0] = c->halo_done_rec[1] = false;
  c->halo_last = -1;
  (void)hipFree(c->d_tile_list);
  (void)hipFree(c->d_tile_state);
  (void)hipFree(c->d_empty_tiles);
  c->d_empty_tiles = nullptr;
  c->empty_tiles_cap = 0;
  c->tile_states_kept = false;
  c->d_tile_list = c->d_tile_state = nullptr;
  (void)hipFree(c->d_counters);
  (void)hipFree(c->d_ids);
  (void)hipFree(c->d_mask);
  (void)hipFree(c->d_brick_tab);
  c->d_brick_tab = nullptr;
  free_lut_arena(c);  // (api_calib.cpp: a plain allocation or a range of mapped chunks)
  (void)hipFree(c->d_win);
  (void)hipFree(c->d_bgmax);
  (void)hipFree(c->d_skip_mask);
  (void)hipFree(c->d_skip_list);
  c->d_skip_list = nullptr;
  c->d_bgmax = nullptr;
  c->d_skip_mask = nullptr;
  c->skip_mask_tiles = 0;
  c->d_win = nullptr;
  c->d_tsdf_base = c->d_tsdf_owned = c->d_linear = nullptr;
  c->d_counters = c->d_ids = nullptr;
  c->d_mask = nullptr;
  c->d_lut_tiled = nullptr;
  c->linear_floats = 0;
  for (int i = 0; i < kMaxSensors; ++i) {
    (void)hipFree(c->d_lut_generic[i]);
    c->d_lut_generic[i] = nullptr;
    c->inv_set[i] = c->inv_tiled[i] = c->inv_resampled[i] = false;
  }
}

// brick table of `g`: counters, id list, two occupied masks and the membership tables.  The new
// buffers are allocated before the old ones are released, so a failure leaves the context as it was.
static int alloc_brick_table(rgbdr_ctx* ctx, const rgbdr_config& cfg, const rgbdr_geometry& g)
{
  BrickTables bt;
  std::string e;
  int rc = compute_brick_tables(cfg, g, &bt, &e);
  if (rc != RGBDR_OK) return ctx->fail(rc, e);
  std::vector<uint32_t> host;
  for (int a = 0; a < 3; ++a) host.insert(host.end(), bt.vox[a].begin(), bt.vox[a].end());
  for (int a = 0; a < 3; ++a) host.insert(host.end(), bt.tile[a].begin(), bt.tile[a].end());
  for (int a = 0; a < 3; ++a) host.insert(host.end(), bt.whole[a].begin(), bt.whole[a].end());
  uint32_t *counters = nullptr, *ids = nullptr, *tab = nullptr;
  uint8_t* mask = nullptr;
  const size_t nb = (size_t)g.num_bricks;
  if (hipMalloc((void**)&counters, 2 * nb * sizeof(uint32_t)) != hipSuccess || hipMalloc((void**)&ids, nb * sizeof(uint32_t)) != hipSuccess ||
      hipMalloc((void**)&mask, nb * 2) != hipSuccess || hipMalloc((void**)&tab, host.size() * sizeof(uint32_t)) != hipSuccess ||
      hipMemcpy(tab, host.data(), host.size() * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemsetAsync(counters, 0, 2 * nb * sizeof(uint32_t), ctx->stream) != hipSuccess ||
      hipMemsetAsync(mask, 0, nb * 2, ctx->stream) != hipSuccess) {
    (void)hipGetLastError();
    (void)hipFree(counters);
    (void)hipFree(ids);
    (void)hipFree(mask);
    (void)hipFree(tab);
    return ctx->fail(RGBDR_ERR_HIP, "allocation of the brick table failed");
  }
  (void)hipFree(ctx->d_counters);
  (void)hipFree(ctx->d_ids);
  (void)hipFree(ctx->d_mask);
  (void)hipFree(ctx->d_brick_tab);
  ctx->d_counters = counters;
  ctx->cbuf = 0;
  ctx->d_ids = ids;
  ctx->d_mask = mask;
  ctx->d_brick_tab = tab;
  ctx->bt = std::move(bt);
  ctx->mask_valid = false;
  ctx->occ_lazy = false;
  return RGBDR_OK;
}

// setVoxelSize / setBrickSize: (re)allocate TSDF slab + brick table
static int alloc_volume(rgbdr_ctx* ctx)
{
  std::string e;
  int rc = compute_geometry(ctx->cfg, &ctx->geo, &e);
  if (rc != RGBDR_OK) return ctx->fail(rc, e);
  free_volume(ctx);
  const rgbdr_geometry& g = ctx->geo;
  ctx->layer_floats = (size_t)g.tiles[0] * g.tiles[1] * kTileVoxels;
  const int owned = g.slab_tile_z1 - g.slab_tile_z0;
  ctx->halo = g.halo_tile_layers;
  const size_t total = ctx->layer_floats * (size_t)(owned + 2 * ctx->halo);
  HIPCHK(hipMalloc((void**)&ctx->d_tsdf_base, total * sizeof(float)));
  ctx->d_tsdf_owned = ctx->d_tsdf_base + ctx->layer_floats * ctx->halo;
  HIPCHK(hipMemsetAsync(ctx->d_tsdf_base, 0, total * sizeof(float), ctx->stream));
  HIPCHK(hipMalloc((void**)&ctx->d_tile_list, ((size_t)g.tiles[0] * g.tiles[1] * owned + 2) * sizeof(uint32_t)));
  HIPCHK(hipMemsetAsync(ctx->d_tile_list + (size_t)g.tiles[0] * g.tiles[1] * owned, 0, 2 * sizeof(uint32_t), ctx->stream));
  ctx->tile_count_parity = 0;
  HIPCHK(hipMalloc((void**)&ctx->d_tile_state, (size_t)g.tiles[0] * g.tiles[1] * owned * sizeof(uint32_t)));
  HIPCHK(hipMemsetAsync(ctx->d_tile_state, 0, (size_t)g.tiles[0] * g.tiles[1] * owned * sizeof(uint32_t), ctx->stream));
  ctx->clear_epoch = 1;
  return alloc_brick_table(ctx, ctx->cfg, g);
}

// a gather rgbdr_shard_allgather_async left on the context's gather stream: whatever touches the frame next on `st` waits
int join_async_gather(rgbdr_ctx* ctx, hipStream_t st)
{
  if (ctx->gather_done_rec) HIPCHK(hipStreamWaitEvent(st, ctx->ev_gather_done, 0));
  return RGBDR_OK;
}

// drain both streams (readbacks, setters, resizes)
int sync_all(rgbdr_ctx* ctx)
{
  HIPCHK(hipSetDevice(ctx->device));
  if (ctx->gather_stream) HIPCHK(hipStreamSynchronize(ctx->gather_stream));
  if (ctx->copy_stream) HIPCHK(hipStreamSynchronize(ctx->copy_stream));
  if (ctx->pre_stream) HIPCHK(hipStreamSynchronize(ctx->pre_stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (ctx->fill_stream) HIPCHK(hipStreamSynchronize(ctx->fill_stream));
  ctx->fill_side = false;
  if (ctx->halo_stream) HIPCHK(hipStreamSynchronize(ctx->halo_stream));
  return RGBDR_OK;
}

}  // namespace rgbdr

namespace rgbdr {
int contain_exception(rgbdr_ctx* ctx) noexcept
{
  int code = RGBDR_ERR_STATE;
  const char* what = "a C++ exception of unknown type was contained at the C boundary";
  char buf[256];
  try {
    throw;
  } catch (const std::bad_alloc&) {
    code = RGBDR_ERR_NO_MEMORY;
    what = "host memory exhausted (std::bad_alloc contained at the C boundary)";
  } catch (const std::length_error&) {
    code = RGBDR_ERR_NO_MEMORY;
    what = "host table larger than a container can hold (std::length_error contained at the C boundary)";
  } catch (const std::exception& e) {
    std::snprintf(buf, sizeof buf, "C++ exception contained at the C boundary: %s", e.what());
    what = buf;
  } catch (...) {
  }
  try {
    (ctx ? ctx->err : g_create_error) = what;
  } catch (...) {  // not even the message fits: the status code still says what happened
  }
  return code;
}
}  // namespace rgbdr

extern "C" {

const char* rgbdr_version(void) { return "rgbdr-hip 0.1 (gfx950)"; }

const char* rgbdr_status_string(int s)
{
  switch (s) {
    case RGBDR_OK: return "ok";
    case RGBDR_ERR_INVALID_ARGUMENT: return "invalid argument";
    case RGBDR_ERR_OUT_OF_RANGE: return "out of range";
    case RGBDR_ERR_NO_DEVICE: return "no HIP device";
    case RGBDR_ERR_HIP: return "HIP runtime error";
    case RGBDR_ERR_IO: return "I/O error";
    case RGBDR_ERR_STATE: return "call order violated";
    case RGBDR_ERR_NO_MEMORY: return "host memory exhausted";
    default: return "unknown status";
  }
}

const char* rgbdr_last_error(const rgbdr_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int rgbdr_compute_geometry(const rgbdr_config* cfg, rgbdr_geometry* out)
try {
  if (!cfg || !out) return RGBDR_ERR_INVALID_ARGUMENT;
  return compute_geometry(*cfg, out, &g_create_error);
}
RGBDR_CONTAIN(nullptr)

int rgbdr_brick_voxel_range(const rgbdr_config* cfg, int axis, int brick, int32_t* first, int32_t* last)
try {
  if (!cfg || !first || !last || axis < 0 || axis > 2) return RGBDR_ERR_INVALID_ARGUMENT;
  rgbdr_geometry g;
  int rc = compute_geometry(*cfg, &g, &g_create_error);
  if (rc != RGBDR_OK) return rc;
  BrickTables bt;
  rc = compute_brick_tables(*cfg, g, &bt, &g_create_error);
  if (rc != RGBDR_OK) return rc;
  if (brick < 0 || brick >= (int)bt.first[axis].size()) return RGBDR_ERR_OUT_OF_RANGE;
  *first = bt.first[axis][brick];
  *last = bt.last[axis][brick];
  return RGBDR_OK;
}
RGBDR_CONTAIN(nullptr)

int rgbdr_slab_range(int tiles_z, int count, int rank, int* t0, int* t1)
try {
  if (!t0 || !t1) return RGBDR_ERR_INVALID_ARGUMENT;
  return slab_range(tiles_z, count, rank, t0, t1);
}
RGBDR_CONTAIN(nullptr)

int rgbdr_camera_position(const rgbdr_lut* cv_xyz, float out[3])
try {
  if (!cv_xyz || !cv_xyz->data || !out) return RGBDR_ERR_INVALID_ARGUMENT;
  if (cv_xyz->res[0] < 1 || cv_xyz->res[1] < 1 || cv_xyz->res[2] < 1) return RGBDR_ERR_INVALID_ARGUMENT;
  camera_position((const float*)cv_xyz->data, cv_xyz->res, out);
  return RGBDR_OK;
}
RGBDR_CONTAIN(nullptr)

int rgbdr_create(const rgbdr_config* cfg, int device_id, rgbdr_ctx** out)
try {
  auto bad = [&](int code, const std::string& m) {
    g_create_error = m;
    return code;
  };
  if (!cfg || !out) return bad(RGBDR_ERR_INVALID_ARGUMENT, "null argument");
  *out = nullptr;
  if (cfg->struct_size != sizeof(rgbdr_config)) return bad(RGBDR_ERR_INVALID_ARGUMENT, "rgbdr_config.struct_size mismatch");
  if (cfg->num_sensors < 1 || cfg->num_sensors > kMaxSensors)
    return bad(RGBDR_ERR_INVALID_ARGUMENT, "num_sensors must be in [1, 8]");
  if (cfg->depth_w < 1 || cfg->depth_h < 1 || cfg->color_w < 1 || cfg->color_h < 1)
    return bad(RGBDR_ERR_INVALID_ARGUMENT, "image sizes must be positive");
  if (cfg->depth_w > 32768 || cfg->depth_h > 32768 || cfg->color_w > 32768 || cfg->color_h > 32768)  // (pixel counts times bytes cannot wrap)
    return bad(RGBDR_ERR_INVALID_ARGUMENT, "image sizes must not exceed 32768 pixels each way");
  if (!(cfg->tsdf_limit >= kMinTsdfLimit) || !std::isfinite(cfg->tsdf_limit))
    return bad(RGBDR_ERR_INVALID_ARGUMENT, "tsdf_limit must be a finite number >= 1e-6 (the ray-marcher steps limit / 2 through the unit cube)");
  if (cfg->compress_rgb != 0 && cfg->compress_rgb != 1 && cfg->compress_rgb != 5)
    return bad(RGBDR_ERR_INVALID_ARGUMENT, "compress_rgb must be 0 (RGB8), 1 (DXT1) or 5 (DXT5)");
  rgbdr_geometry g;
  std::string e;
  int rc = compute_geometry(*cfg, &g, &e);
  if (rc != RGBDR_OK) return bad(rc, e);

  int ndev = 0;
  hipError_t he = hipGetDeviceCount(&ndev);
  if (he != hipSuccess || ndev < 1)
    return bad(RGBDR_ERR_NO_DEVICE, std::string("no HIP device available (") + hipGetErrorString(he) +
                                        "); this backend has no CPU fallback");
  if (device_id < 0 || device_id >= ndev) return bad(RGBDR_ERR_OUT_OF_RANGE, "device_id out of range");
  if (hipSetDevice(device_id) != hipSuccess) return bad(RGBDR_ERR_NO_DEVICE, "hipSetDevice failed");

  rgbdr_ctx* ctx = new rgbdr_ctx();
  struct Unwind {  // an exception below must not leak the half-made context (the handler runs after the locals are gone)
    rgbdr_ctx* c;
    ~Unwind() { if (c) rgbdr_destroy(c); }
  } unwind{ctx};
  ctx->cfg = *cfg;
  if (ctx->cfg.slab_count <= 0) {
    ctx->cfg.slab_count = 1;
    ctx->cfg.slab_rank = 0;
  }
  ctx->device = device_id;
  auto cleanup = [&](int code) {
    g_create_error = ctx->err;
    unwind.c = nullptr;
    rgbdr_destroy(ctx);
    return code;
  };
  // Developer experiment (round 4): RGBDR_CU_SPLIT=n gives the second stream (pre_* chain of the two-stream schedule)
  // n compute units of its own and keeps the sweep's stream off them (hipExtStreamCreateWithCUMask).  Without it the chain's 256-thread blocks
  // starve next to a sweep whose 262 144 blocks refill every slot that frees up (the chain runs 8-20x longer there).
  int cu_split = 0;
  if (const char* env = std::getenv("RGBDR_CU_SPLIT")) cu_split = std::atoi(env);
  hipDeviceProp_t prop{};
  if (cu_split > 0 && (hipGetDeviceProperties(&prop, device_id) != hipSuccess || cu_split >= prop.multiProcessorCount)) cu_split = 0;
  bool masked = false;
  if (cu_split > 0) {
    const int ncu = prop.multiProcessorCount, words = (ncu + 31) / 32;
    std::vector<uint32_t> side((size_t)words, 0u), mainm((size_t)words, 0u);
    for (int i = 0; i < ncu; ++i) {
      // KFD hands mask bit i to XCC i mod 8 (and spreads an XCC's bits over its shader engines), so a CONTIGUOUS run of
      // n bits takes n / 8 CUs from every XCD; bits 0, 32, 64 ... would all come out of XCD 0, whose quarter of the
      // statically round-robined workgroups then runs 25 % longer (measured: sweep 1.06 -> 1.28 ms with 8 such CUs)
      const bool reserved = i < cu_split;
      (reserved ? side : mainm)[(size_t)i / 32] |= 1u << (i % 32);
    }
    masked = hipExtStreamCreateWithCUMask(&ctx->own_stream, (uint32_t)words, mainm.data()) == hipSuccess &&
             hipExtStreamCreateWithCUMask(&ctx->pre_stream, (uint32_t)words, side.data()) == hipSuccess;
    if (masked) ctx->side_cu_mask = side;
    if (!masked) {
      (void)hipGetLastError();
      if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
      if (ctx->pre_stream) (void)hipStreamDestroy(ctx->pre_stream);
      ctx->own_stream = ctx->pre_stream = nullptr;
      std::fprintf(stderr, "rgbdr: RGBDR_CU_SPLIT=%d: hipExtStreamCreateWithCUMask failed, using plain streams\n", cu_split);
    }
  }
  if (!masked && hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) {
    ctx->err = "hipStreamCreate failed";
    return cleanup(RGBDR_ERR_HIP);
  }
  ctx->stream = ctx->own_stream;
  if (!masked) {
    // the small pre_* kernels of the next frame must not queue behind the 262 144 workgroups
    // of an integrate sweep: give their stream the highest dispatch priority
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    if (hipStreamCreateWithPriority(&ctx->pre_stream, hipStreamNonBlocking, greatest) != hipSuccess) {
      ctx->err = "hipStreamCreate failed";
      return cleanup(RGBDR_ERR_HIP);
    }
  }
  if (hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking) != hipSuccess) {
    ctx->err = "hipStreamCreate failed";
    return cleanup(RGBDR_ERR_HIP);
  }
  // The events that order this context's own streams against each other (kernels of one device on both sides) need no
  // system-scope fence: with it a record writes the caches back for the host's and other devices' benefit and costs the
  // stream ~20 us (profiles/r06_notes/display_pipeline.md).  RGBDR_DEV_SYSTEM_FENCE=1: the default fence, for A/B runs.
  ctx->fenceless_events = cfg->slab_count <= 1 && !std::getenv("RGBDR_DEV_SYSTEM_FENCE");  // (a slab's halos come from other devices)
  const unsigned own = hipEventDisableTiming | (ctx->fenceless_events ? hipEventDisableSystemFence : 0u);
  for (int b = 0; b < 2; ++b)
    if (hipEventCreateWithFlags(&ctx->ev_pre[b], own) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_h2d[b], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_in_read[b], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_int[b], own) != hipSuccess) {
      ctx->err = "hipEventCreate failed";
      return cleanup(RGBDR_ERR_HIP);
    }
  for (int b = 0; b < 2; ++b)
    if (hipEventCreateWithFlags(&ctx->ev_view_read[b], own) != hipSuccess) {
      ctx->err = "hipEventCreate failed";
      return cleanup(RGBDR_ERR_HIP);
    }
  const size_t n = npx(ctx);
  const size_t ncol = (size_t)cfg->num_sensors * cfg->color_w * cfg->color_h * 3;
  struct {
    void** p;
    size_t bytes;
  } allocs[] = {{(void**)&ctx->d_depth_raw, n * 4},   {(void**)&ctx->d_depth_morph, n * 4},
                {(void**)&ctx->d_depth_rg, n * 8},    {(void**)&ctx->d_lab, n * 12},
                {(void**)&ctx->d_depth_b, n * 8},     {(void**)&ctx->d_sil, n * 4},
                {(void**)&ctx->d_normal, n * 12},     {(void**)&ctx->d_quality, n * 4},
                {(void**)&ctx->d_frame, n * 8 * 2},      {(void**)&ctx->d_color, ncol * 2},
                {(void**)&ctx->d_count, 32},
                {(void**)&ctx->d_color_dxt, color_frame_bytes(*cfg) * cfg->num_sensors * 2},
                {(void**)&ctx->d_cc_far, n * 8},         {(void**)&ctx->d_box_flags, n}};
  ctx->color_half_bytes = ncol;
  ctx->dxt_half_bytes = color_frame_bytes(*cfg) * cfg->num_sensors;
  for (auto& a : allocs) {
    if (hipMalloc(a.p, a.bytes) != hipSuccess) {
      ctx->err = "hipMalloc of image buffers failed";
      return cleanup(RGBDR_ERR_HIP);
    }
    (void)hipMemsetAsync(*a.p, 0, a.bytes, ctx->stream);
  }
  // 13x13 spatial kernel of the bilateral filter, pre_depth.fs:37-41,115
  float gauss[169];
  const float inv_k = 1.0f / 6.0f;
  for (int y = -6; y < 7; ++y)
    for (int x = -6; x < 7; ++x) {
      const float len = std::sqrt((float)x * (float)x + (float)y * (float)y);
      gauss[(y + 6) * 13 + (x + 6)] = 1.0f - len * inv_k;
    }
  set_gauss_table(gauss);
  rc = alloc_volume(ctx);
  if (rc != RGBDR_OK) return cleanup(rc);
  if (hipStreamSynchronize(ctx->stream) != hipSuccess) {
    ctx->err = "device initialisation failed";
    return cleanup(RGBDR_ERR_HIP);
  }
  unwind.c = nullptr;
  *out = ctx;
  return RGBDR_OK;
}
RGBDR_CONTAIN(nullptr)

void rgbdr_destroy(rgbdr_ctx* ctx)
{
  if (!ctx) return;
  // drain every stream first: queued kernels still write the mapped skip counter and read the page-locked frame buffers
  (void)hipSetDevice(ctx->device);
  rgbdr::release_peer_waits(ctx, -1);  // (a copy-engine halo step may still wait for a neighbour that is gone)
  if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
  if (ctx->pre_stream) (void)hipStreamSynchronize(ctx->pre_stream);
  if (ctx->halo_stream) (void)hipStreamSynchronize(ctx->halo_stream);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  for (int b = 0; b < 2; ++b) {
    (void)hipFree(ctx->d_in_depth[b]);
    (void)hipFree(ctx->d_in_color[b]);
    if (ctx->ev_h2d[b]) (void)hipEventDestroy(ctx->ev_h2d[b]);
    if (ctx->ev_in_read[b]) (void)hipEventDestroy(ctx->ev_in_read[b]);
    if (ctx->h_depth[b]) (void)hipHostFree(ctx->h_depth[b]);
    if (ctx->h_color[b]) (void)hipHostFree(ctx->h_color[b]);
    if (b == 0 && ctx->h_skip_count) (void)hipHostFree(ctx->h_skip_count);
    if (ctx->ev_mapped[b]) (void)hipEventDestroy(ctx->ev_mapped[b]);
    ctx->h_depth[b] = ctx->h_color[b] = nullptr;
    ctx->ev_mapped[b] = nullptr;
  }
  free_volume(ctx);
  for (int b = 0; b < 2; ++b) {
    if (ctx->ev_pre[b]) (void)hipEventDestroy(ctx->ev_pre[b]);
    if (ctx->ev_int[b]) (void)hipEventDestroy(ctx->ev_int[b]);
  }
  for (hipEvent_t e : ctx->ev_view_read)
    if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : ctx->ev_fill)
    if (e) (void)hipEventDestroy(e);
  if (ctx->ev_view_ready) (void)hipEventDestroy(ctx->ev_view_ready);
  if (ctx->fill_stream) (void)hipStreamDestroy(ctx->fill_stream);
  if (ctx->pre_stream) (void)hipStreamDestroy(ctx->pre_stream);
  if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
  if (ctx->gather_stream) {
    (void)hipStreamSynchronize(ctx->gather_stream);
    (void)hipStreamDestroy(ctx->gather_stream);
  }
  for (hipEvent_t e : {ctx->ev_gather_from, ctx->ev_gather_done, ctx->ev_export, ctx->ev_imported})
    if (e) (void)hipEventDestroy(e);
  if (ctx->halo_stream) {
    (void)hipStreamSynchronize(ctx->halo_stream);
    (void)hipStreamDestroy(ctx->halo_stream);
  }
  for (int b = 0; b < 2; ++b) {
    if (ctx->ev_halo_staged[b]) (void)hipEventDestroy(ctx->ev_halo_staged[b]);
    if (ctx->ev_halo_done[b]) (void)hipEventDestroy(ctx->ev_halo_done[b]);
  }
  void* ptrs[] = {ctx->d_depth_raw, ctx->d_depth_morph, ctx->d_depth_rg, ctx->d_lab,   ctx->d_depth_b, ctx->d_sil,
                  ctx->d_normal,    ctx->d_quality,     ctx->d_frame,    ctx->d_color, ctx->d_count,
                  ctx->d_color_dxt, ctx->d_cc_far,    ctx->d_box_flags};
  for (void* p : ptrs) (void)hipFree(p);
  for (int i = 0; i < kMaxSensors; ++i) {
    (void)hipFree(ctx->d_cv_xyz[i]);
    (void)hipFree(ctx->d_cv_uv[i]);
  }
  destroy_timers(ctx);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
  delete ctx;
}

// ---------------------------------------------------------------------------
static size_t depth_frame_bytes_all(const rgbdr_ctx* ctx) { return npx(ctx) * (ctx->cfg.compress_depth ? 1 : 4); }
static size_t color_frame_bytes_all(const rgbdr_ctx* ctx) { return color_frame_bytes(ctx->cfg) * (size_t)nsens(ctx); }

// The frame set is on the device (the caller's buffers, or a staging set a host upload filled): raw depth (or u8
// depth), the pre_morph image and the colour frame / blocks in as few launches as the pointers allow, on the stream
// that runs the pre_* chain.
static int upload_device(rgbdr_ctx* ctx, const void* depth, const void* color)
{
  const size_t n = npx(ctx);
  const size_t ncol = (size_t)nsens(ctx) * ctx->cfg.color_w * ctx->cfg.color_h * 3;
  hipStream_t ps = ctx->pstream();
  // The colour half this frame goes to: on a pipelined context the one the frame before does not live in (its view pass,
  // rgbdr_draw on the other stream, may still read that; a zero-copy view of the colour image pins the half it was given
  // for); a frame no pre_* chain has looked at is simply replaced.  Whoever read the target last has to be done with it.
  int ch = ctx->color_up;
  if (ctx->pipelined() && !ctx->color_view_out && ctx->color_consumed) ch ^= 1;
  for (int b = 0; b < 2; ++b)
    if (ctx->pipelined() && ctx->ev_view_rec[b] && ctx->view_color[b] == ch) HIPCHK(hipStreamWaitEvent(ps, ctx->ev_view_read[b], 0));
  ctx->color_up = ch;
  ctx->color_consumed = false;
  uint8_t *const d_color = ctx->color_half(ch), *const d_color_dxt = ctx->dxt_half(ch);
  ctx->morph_current = false;
  ctx->frame_uploaded = false;  // until every launch below is enqueued: an upload that fails part-way leaves no frame
  // a sensor shard (rgbdr_set_sensor_shard) copies and morphs the raw depth of its own layers only
  const int first = ctx->shard_count > 0 ? ctx->shard_first : 0, count = ctx->shard_count > 0 ? ctx->shard_count : nsens(ctx);
  if (!ctx->cfg.compress_depth && !ctx->cfg.compress_rgb &&
      launch_upload_morph(ctx->cfg.depth_w, ctx->cfg.depth_h, first, count, depth, ctx->d_depth_raw, ctx->d_depth_morph, color,
                          d_color, ncol, ps)) {
    LAUNCHCHK("upload_morph");
    ctx->morph_current = true;
    ctx->color_decoded[ch] = true;
    ctx->frame_uploaded = true;
    return RGBDR_OK;
  }
  // Compressed colour stays in its DXT blocks: pre_depth decodes the four taps of its bilinear lookup itself, and the
  // whole frame is only decoded when a consumer of the RGB8 image asks (ensure_color_decoded).
  const size_t layer = color_frame_bytes(ctx->cfg);
  void* cdst = ctx->cfg.compress_rgb ? (void*)d_color_dxt : (void*)d_color;
  const size_t cbytes = ctx->cfg.compress_rgb ? layer * nsens(ctx) : ncol;
  bool color_done = false;
  if (ctx->cfg.compress_depth) {
    launch_u8_to_unit((const uint8_t*)depth, ctx->d_depth_raw, n, ps);
    LAUNCHCHK("u8_to_unit");
  } else if (launch_upload_morph(ctx->cfg.depth_w, ctx->cfg.depth_h, first, count, depth, ctx->d_depth_raw, ctx->d_depth_morph,
                                 color, cdst, cbytes, ps)) {
    color_done = true;
    ctx->morph_current = true;
  } else {
    HIPCHK(hipMemcpyAsync(ctx->d_depth_raw, depth, n * 4, hipMemcpyDeviceToDevice, ps));
  }
  if (!color_done) HIPCHK(hipMemcpyAsync(cdst, color, cbytes, hipMemcpyDeviceToDevice, ps));
  ctx->color_decoded[ch] = !ctx->cfg.compress_rgb;
  ctx->frame_uploaded = true;
  // a zero-copy view of the RGB8 frame is out (rgbdr_device_image(RGBDR_IMG_COLOR)): it is documented as rewritten
  // by every upload, so the decode rides behind the copy on the same stream -- no host synchronisation
  if (!ctx->color_decoded[ch] && ctx->color_view_out) {
    launch_decode_dxt(d_color_dxt, ctx->cfg.color_w, ctx->cfg.color_h, ctx->cfg.compress_rgb, nsens(ctx),
                      color_frame_bytes(ctx->cfg), d_color, ps);
    LAUNCHCHK("decode_dxt");
    ctx->color_decoded[ch] = true;
  }
  return RGBDR_OK;
}

// NetKinectArray::update (NetKinectArray.cpp:226-238: PBO -> texture arrays) for frames in HOST memory.  The copy has
// its own stream and lands in a staging set, so the DMA of frame k+1 runs while the sweep of frame k still occupies
// the compute stream; the chain's stream only waits for the event behind the copy and then runs the same launch a
// device-resident frame takes.  Page-locked sources (the double frame buffer) are true asynchronous DMA; a pageable
// source is staged by the runtime before hipMemcpyAsync returns, so the caller may reuse its buffer at once either way.
static int upload_host(rgbdr_ctx* ctx, const void* depth, const void* color, bool caller_buffers)
{
  const size_t dbytes = depth_frame_bytes_all(ctx), cbytes = color_frame_bytes_all(ctx);
  const int s = ctx->in_set;
  if (!ctx->d_in_depth[s]) HIPCHK(hipMalloc(&ctx->d_in_depth[s], dbytes));  // (checked one by one: a failure of the
  if (!ctx->d_in_color[s]) HIPCHK(hipMalloc(&ctx->d_in_color[s], cbytes));  // second must not leave a half-made set)
  hipStream_t cs = ctx->copy_stream;
  if (ctx->ev_in_read_rec[s]) HIPCHK(hipStreamWaitEvent(cs, ctx->ev_in_read[s], 0));  // its last reader: two uploads ago
  HIPCHK(hipMemcpyAsync(ctx->d_in_depth[s], depth, dbytes, hipMemcpyHostToDevice, cs));
  HIPCHK(hipMemcpyAsync(ctx->d_in_color[s], color, cbytes, hipMemcpyHostToDevice, cs));
  HIPCHK(hipEventRecord(ctx->ev_h2d[s], cs));
  // rgbdr_upload_frame's contract: the caller may reuse its buffers when the call returns.  A pageable source has been
  // staged by then; a source the caller page-locked itself is read by the DMA engine asynchronously, so wait for the
  // copy (not for anything else: the copy stream runs ahead of the passes).  The library's own double buffer is handed
  // back through rgbdr_map_frame_buffer, which waits for the same event.
  if (caller_buffers) HIPCHK(hipEventSynchronize(ctx->ev_h2d[s]));
  hipStream_t ps = ctx->pstream();
  HIPCHK(hipStreamWaitEvent(ps, ctx->ev_h2d[s], 0));
  const int rc = upload_device(ctx, ctx->d_in_depth[s], ctx->d_in_color[s]);
  // Whatever upload_device enqueued on `ps` before it failed may still read staging set s: the event is recorded on
  // the failure path too, so the next host upload (which DMAs into a staging set on the copy stream) waits for it.
  const hipError_t ev = hipEventRecord(ctx->ev_in_read[s], ps);
  if (ev == hipSuccess) ctx->ev_in_read_rec[s] = true;
  else (void)hipStreamSynchronize(ps);  // no event to order against: drain the readers now
  if (rc != RGBDR_OK) return rc;
  HIPCHK(ev);
  ctx->in_set = 1 - s;
  return RGBDR_OK;
}

static int upload_common(rgbdr_ctx* ctx, const void* depth, const void* color, hipMemcpyKind kind, bool caller_buffers = true)
{
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (!depth || !color) return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, "null frame pointer");
  HIPCHK(hipSetDevice(ctx->device));
  return kind == hipMemcpyDeviceToDevice ? upload_device(ctx, depth, color) : upload_host(ctx, depth, color, caller_buffers);
}

int rgbdr_upload_frame(rgbdr_ctx* ctx, const void* depth, const void* color)
try {
  return upload_common(ctx, depth, color, hipMemcpyHostToDevice);
}
RGBDR_CONTAIN(ctx)
int rgbdr_upload_frame_device(rgbdr_ctx* ctx, const void* depth, const void* color)
try {
  return upload_common(ctx, depth, color, hipMemcpyDeviceToDevice);
}
RGBDR_CONTAIN(ctx)

int rgbdr_map_frame_buffer(rgbdr_ctx* ctx, void** depth, void** color, size_t* depth_bytes, size_t* color_bytes)
try {
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (!depth || !color) return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, "null output pointer");
  HIPCHK(hipSetDevice(ctx->device));
  const int b = ctx->mapped_back;
  if (!ctx->h_depth[b]) {
    HIPCHK(hipHostMalloc(&ctx->h_depth[b], depth_frame_bytes_all(ctx), hipHostMallocDefault));
    HIPCHK(hipHostMalloc(&ctx->h_color[b], color_frame_bytes_all(ctx), hipHostMallocDefault));
    HIPCHK(hipEventCreateWithFlags(&ctx->ev_mapped[b], hipEventDisableTiming));
  }
  // the copy that last read this buffer (two uploads ago) must have drained before it is refilled
  if (ctx->ev_mapped_rec[b]) HIPCHK(hipEventSynchronize(ctx->ev_mapped[b]));
  *depth = ctx->h_depth[b];
  *color = ctx->h_color[b];
  if (depth_bytes) *depth_bytes = depth_frame_bytes_all(ctx);
  if (color_bytes) *color_bytes = color_frame_bytes_all(ctx);
  return RGBDR_OK;
}
RGBDR_CONTAIN(ctx)

int rgbdr_upload_mapped_frame(rgbdr_ctx* ctx)
try {
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  const int b = ctx->mapped_back;
  if (!ctx->h_depth[b]) return ctx->fail(RGBDR_ERR_STATE, "upload_mapped_frame before map_frame_buffer");
  int rc = upload_common(ctx, ctx->h_depth[b], ctx->h_color[b], hipMemcpyHostToDevice, false);  // page-locked: true async DMA
  if (rc != RGBDR_OK) return rc;
  HIPCHK(hipEventRecord(ctx->ev_mapped[b], ctx->copy_stream));  // the DMA out of this buffer has drained
  ctx->ev_mapped_rec[b] = true;
  ctx->mapped_back = 1 - b;  // swapBuffers
  return RGBDR_OK;
}
RGBDR_CONTAIN(ctx)

int rgbdr_clear_occupied_bricks(rgbdr_ctx* ctx)
try {
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (!ctx->clear_pending) {
    // the counting moves to the other buffer (zeroed by the first kernel of process_textures, or by whoever reads
    // the counters first): a filter still pending on the one just left is not disturbed; one pending on the buffer
    // coming back into use (two clears without an update in between) has to be evaluated now
    if (ctx->occ_lazy && ctx->occ_lazy_cbuf == (ctx->cbuf ^ 1)) {
      int rc_ = materialise_mask(ctx);
      if (rc_ != RGBDR_OK) return rc_;
    }
    ctx->cbuf ^= 1;
  }
  ctx->clear_pending = true;
  return RGBDR_OK;
}
RGBDR_CONTAIN(ctx)

}  // extern "C"
// clearOccupiedBricks is deferred; anything that reads the counters before process_textures ran flushes it
// the RGB8 colour frame for consumers other than pre_depth (which reads DXT blocks directly)
int rgbdr::system_fence_events(rgbdr_ctx* ctx)
{
  if (!ctx->fenceless_events) return RGBDR_OK;
  { int rc_ = sync_all(ctx); if (rc_ != RGBDR_OK) return rc_; }  // nothing is in flight: the old events have nothing left to say
  for (int b = 0; b < 2; ++b) {
    hipEvent_t* evs[3] = {&ctx->ev_pre[b], &ctx->ev_int[b], &ctx->ev_view_read[b]};
    for (hipEvent_t* e : evs) {
      hipEvent_t fresh = nullptr;
      HIPCHK(hipEventCreateWithFlags(&fresh, hipEventDisableTiming));
      if (*e) (void)hipEventDestroy(*e);
      *e = fresh;
    }
    ctx->ev_pre_rec[b] = ctx->ev_int_rec[b] = ctx->ev_view_rec[b] = ctx->int_unrecorded[b] = false;
  }
  ctx->draw_expected = false;
  ctx->fenceless_events = false;
  return RGBDR_OK;
}
// the sweep and the view pass of the frame that lives in half w of the frame buffers (images, mask, counts) read it on the
// first stream: the chain's stream waits for them before it writes there
int rgbdr::wait_last_readers(rgbdr_ctx* ctx, int w, hipStream_t ps)
{
  if (!ctx->pipelined()) return RGBDR_OK;
  if (ctx->int_unrecorded[w]) {  // its sweep left the record to a draw that did not come: behind what the first stream holds now
    HIPCHK(hipEventRecord(ctx->ev_int[w], ctx->stream));
    ctx->ev_int_rec[w] = true;
    ctx->int_unrecorded[w] = false;
    ctx->draw_expected = false;
  }
  if (ctx->ev_int_rec[w]) HIPCHK(hipStreamWaitEvent(ps, ctx->ev_int[w], 0));
  if (ctx->ev_view_rec[w]) HIPCHK(hipStreamWaitEvent(ps, ctx->ev_view_read[w], 0));
  return RGBDR_OK;
}
int rgbdr::ensure_color_decoded(rgbdr_ctx* ctx, int half)
{
  const int h = half < 0 ? ctx->color_up : half;  // (the image getters: the last uploaded frame)
  if (ctx->color_decoded[h]) return RGBDR_OK;
  HIPCHK(hipSetDevice(ctx->device));
  launch_decode_dxt(ctx->dxt_half(h), ctx->cfg.color_w, ctx->cfg.color_h, ctx->cfg.compress_rgb, nsens(ctx),
                    color_frame_bytes(ctx->cfg), ctx->color_half(h), ctx->pstream());
  LAUNCHCHK("decode_dxt");
  if (ctx->pipelined()) HIPCHK(hipStreamSynchronize(ctx->pstream()));  // consumers may sit on the other stream
  ctx->color_decoded[h] = true;
  return RGBDR_OK;
}
// the filter of a lazy rgbdr_update_occupied_bricks, for consumers other than the brick sweep and before
// anything changes the counters it was asked for
int rgbdr::materialise_mask(rgbdr_ctx* ctx)
{
  if (!ctx->occ_lazy) return RGBDR_OK;
  HIPCHK(hipSetDevice(ctx->device));
  launch_update_occupied(ctx->d_counters + (size_t)ctx->occ_lazy_cbuf * ctx->geo.num_bricks, (uint32_t)ctx->geo.num_bricks,
                         ctx->occ_lazy_min, ctx->mask_buf(ctx->rbuf),
                         ctx->count_buf(ctx->rbuf), ctx->pstream());
  LAUNCHCHK("update_occupied");
  ctx->occ_lazy = false;
  return RGBDR_OK;
}
int rgbdr::flush_clear(rgbdr_ctx* ctx)
{
  if (!ctx->clear_pending) return RGBDR_OK;
  if (ctx->occ_lazy && ctx->occ_lazy_cbuf == ctx->cbuf) { int rc_ = materialise_mask(ctx); if (rc_ != RGBDR_OK) return rc_; }
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipMemsetAsync(ctx->counters_cur(), 0, (size_t)ctx->geo.num_bricks * sizeof(uint32_t), ctx->pstream()));
  ctx->clear_pending = false;
  return RGBDR_OK;
}
extern "C" {

int rgbdr_process_textures(rgbdr_ctx* ctx)
try {
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (!ctx->frame_uploaded) return ctx->fail(RGBDR_ERR_STATE, "process_textures before any frame was uploaded");
  for (int i = 0; i < nsens(ctx); ++i)
    if (!ctx->have_calib[i]) return ctx->fail(RGBDR_ERR_STATE, "process_textures before set_calibration of every sensor");
  HIPCHK(hipSetDevice(ctx->device));
  PreParams p{};
  p.N = nsens(ctx);
  p.first = ctx->shard_count > 0 ? ctx->shard_first : 0;
  p.count = ctx->shard_count > 0 ? ctx->shard_count : p.N;
  p.W = ctx->cfg.depth_w;
  p.H = ctx->cfg.depth_h;
  p.Wc = ctx->cfg.color_w;
  p.Hc = ctx->cfg.color_h;
  for (int a = 0; a < 3; ++a) {
    p.bbox_min[a] = ctx->cfg.bbox_min[a];
    p.bbox_max[a] = ctx->cfg.bbox_max[a];
    p.res_bricks[a] = ctx->geo.res_bricks[a];
  }
  p.filter = (ctx->cfg.flags & RGBDR_FLAG_FILTER) ? 1 : 0;
  p.refine = (ctx->cfg.flags & RGBDR_FLAG_REFINE) ? 1 : 0;
  p.compress = ctx->cfg.compress_depth ? 1 : 0;
  for (int i = 0; i < p.N; ++i) {
    p.cv_xyz[i] = ctx->d_cv_xyz[i];
    p.cv_uv[i] = ctx->d_cv_uv[i];
    for (int a = 0; a < 3; ++a) {
      p.xyz_res[i][a] = (int)ctx->xyz_res[i][a];
      p.uv_res[i][a] = (int)ctx->uv_res[i][a];
      p.cam_pos[i][a] = ctx->cam_pos[i][a];
    }
    p.cv_min_ds[i] = ctx->min_ds[i];
    p.cv_max_ds[i] = ctx->max_ds[i];
    p.near_[i] = ctx->cfg.near_[i];
    p.far_[i] = ctx->cfg.far_[i];
  }
  p.brick_size = ctx->geo.brick_size;
  p.brick_counters = ctx->counters_cur();
  p.color = ctx->color_half(ctx->color_up);
  p.color_dxt = ctx->color_decoded[ctx->color_up] ? nullptr : ctx->dxt_half(ctx->color_up);
  p.color_layer_bytes = color_frame_bytes(ctx->cfg);
  p.color_mode = ctx->cfg.compress_rgb;
  p.depth_morph = ctx->d_depth_morph;
  p.depth_rg = ctx->d_depth_rg;
  p.lab = ctx->d_lab;
  p.depth_b_rg = ctx->d_depth_b;
  p.silhouette = ctx->d_sil;
  p.normal = ctx->d_normal;
  p.quality = ctx->d_quality;
  p.cc_far = ctx->d_cc_far;
  p.box_flags = ctx->d_box_flags;
  // m_use_processed_depth: the filter pass reads the morph output instead of the
  // raw depth (NetKinectArray.cpp:287-289)
  p.depth_in = (ctx->cfg.flags & RGBDR_FLAG_PROCESSED) ? ctx->d_depth_morph : ctx->d_depth_raw;

  if (ctx->occ_lazy && ctx->occ_lazy_cbuf == ctx->cbuf) {  // the counters a pending filter reads are about to change
    int rc_ = materialise_mask(ctx);
    if (rc_ != RGBDR_OK) return rc_;
  }
  hipStream_t ps = ctx->pstream();
  const int w = ctx->wbuf;
  p.frame = ctx->frame_buf(w);
  { int rc_ = wait_last_readers(ctx, w, ps); if (rc_ != RGBDR_OK) return rc_; }
  { int rc_ = join_async_gather(ctx, ps); if (rc_ != RGBDR_OK) return rc_; }  // (an asynchronous gather still writes the frame / the counters)
  // another context may still be copying the LAST frame out of these buffers (rgbdr_import_frame_from on a stream of its own)
  if (ctx->imported_rec) HIPCHK(hipStreamWaitEvent(ps, ctx->ev_imported, 0));
  tbegin(ctx, "1preprocess", ps);
  // whichever kernel comes first performs a pending clearOccupiedBricks; the morph image of a frame that was
  // uploaded from device memory was written with the upload
  uint32_t* zero = ctx->clear_pending ? ctx->counters_cur() : nullptr;
  tbegin(ctx, "morph", ps);
  if (!ctx->morph_current) {
    launch_morph(p, ctx->d_depth_raw, ctx->d_depth_morph, zero, (unsigned)ctx->geo.num_bricks, ps);
    zero = nullptr;
  }
  tend(ctx, "morph", ps);
  tbegin(ctx, "bilateral", ps);
  launch_pre_depth(p, zero, (unsigned)ctx->geo.num_bricks, ps);
  ctx->clear_pending = false;
  tend(ctx, "bilateral", ps);
  // the boundary, normal and quality passes run as one launch unless the host asked for the per-pass timers
  // ("boundary" / "normal" / "quality" of NetKinectArray.cpp:359-414), which need the separate ones
  if ((ctx->timers && ctx->timer_detail >= 2) || ctx->separate_passes) {
    tbegin(ctx, "boundary", ps);
    launch_boundary(p, ps);
    tend(ctx, "boundary", ps);
    tbegin(ctx, "normal", ps);
    launch_normal(p, ps);
    tend(ctx, "normal", ps);
    tbegin(ctx, "quality", ps);
    launch_quality(p, ps);
    tend(ctx, "quality", ps);
  } else if (ctx->fuse_boundary) {
    launch_boundary_normal_quality(p, ps);
  } else {
    launch_boundary(p, ps);
    launch_normal_quality(p, ps);
  }
  tend(ctx, "1preprocess", ps);
  LAUNCHCHK("process_textures");
  ctx->rbuf = w;
  ctx->color_of[w] = ctx->color_up;
  ctx->color_consumed = true;
  if (ctx->pipelined()) {
    HIPCHK(hipEventRecord(ctx->ev_pre[w], ps));
    ctx->ev_pre_rec[w] = true;
    ++ctx->pre_serial;
  }
  ctx->textures_processed = true;
  ctx->bgmax_for = -1;
  ctx->shard_pending = p.count < p.N;  // the other sensors' frames and the other ranks' brick counts are still to come
  return RGBDR_OK;
}
RGBDR_CONTAIN(ctx)

int rgbdr_update_occupied_bricks(rgbdr_ctx* ctx)
try {
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (ctx->shard_pending)
    return ctx->fail(RGBDR_ERR_STATE, "update_occupied_bricks on a sensor shard before rgbdr_shard_allgather: the brick counters hold "
                                      "this rank's sensors only");
  HIPCHK(hipSetDevice(ctx->device));
  { int rc_ = flush_clear(ctx); if (rc_ != RGBDR_OK) return rc_; }
  hipStream_t ps = ctx->pstream();
  { int rc_ = join_async_gather(ctx, ps); if (rc_ != RGBDR_OK) return rc_; }
  const int w = ctx->rbuf;  // belongs to the frame process_textures just wrote
  tbegin(ctx, "bricks", ps);
  if (!ctx->pipelined()) {
    // one stream: nothing can touch the counters before the next call of this library does, so the filter is left
    // to the first consumer -- the brick sweep folds it into its first kernel
    ctx->occ_lazy = true;  // (replaces a filter still pending from the frame before: nobody asked for it)
    ctx->occ_lazy_min = ctx->cfg.min_voxels_per_brick;
    ctx->occ_lazy_cbuf = ctx->cbuf;
  } else {
    ctx->occ_lazy = false;
    launch_update_occupied(ctx->counters_cur(), (uint32_t)ctx->geo.num_bricks, ctx->cfg.min_voxels_per_brick,
                           ctx->mask_buf(w), ctx->count_buf(w), ps);
  }
  tend(ctx, "bricks", ps);
  LAUNCHCHK("update_occupied");
  if (ctx->pipelined()) {
    HIPCHK(hipEventRecord(ctx->ev_pre[w], ps));
    ctx->ev_pre_rec[w] = true;
    ++ctx->pre_serial;
  }
  ctx->mask_valid = true;
  return RGBDR_OK;
}
RGBDR_CONTAIN(ctx)

int rgbdr_set_occupied_bricks(rgbdr_ctx* ctx, const uint32_t* ids, size_t count)
try {
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (!ids && count) return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, "null id list");
  const size_t nb = (size_t)ctx->geo.num_bricks;
  std::vector<uint8_t> mask(nb, 0);
  for (size_t i = 0; i < count; ++i) {
    if (ids[i] >= nb) return ctx->fail(RGBDR_ERR_OUT_OF_RANGE, "brick id out of range");
    mask[ids[i]] = 1;
  }
  { int rc_ = sync_all(ctx); if (rc_ != RGBDR_OK) return rc_; }
  HIPCHK(hipMemcpy(ctx->mask_buf(ctx->rbuf), mask.data(), nb, hipMemcpyHostToDevice));
  ctx->occ_lazy = false;  // the host's list replaces whatever the counters would have given
  ctx->mask_valid = true;
  return RGBDR_OK;
}
RGBDR_CONTAIN(ctx)

int rgbdr_integrate(rgbdr_ctx* ctx)
try {
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (!ctx->textures_processed) return ctx->fail(RGBDR_ERR_STATE, "integrate before process_textures");
  if (ctx->shard_pending)
    return ctx->fail(RGBDR_ERR_STATE, "integrate on a sensor shard before rgbdr_shard_allgather: the frames of the other ranks' sensors "
                                      "have not arrived");
  const int N = nsens(ctx);
  bool all_tiled = true, any_tiled = false;
  for (int i = 0; i < N; ++i) {
    if (!ctx->inv_set[i])
      return ctx->fail(RGBDR_ERR_STATE, "integrate before set_inverse_calibration of every sensor (it must be repeated "
                                        "after the grid was resized)");
    all_tiled = all_tiled && ctx->inv_tiled[i];
    any_tiled = any_tiled || ctx->inv_tiled[i];
  }
  if (any_tiled && !all_tiled)
    // unreachable through the API: every call that sets an inverse LUT keeps the sensors of a context in one layout
    // (any mix of LUT resolutions is fine in either layout)
    return ctx->fail(RGBDR_ERR_STATE, "internal: sensors hold inverse LUTs in different layouts");
  const bool bricks = (ctx->cfg.flags & RGBDR_FLAG_USE_BRICKS) != 0;
  if (bricks && !ctx->mask_valid) return ctx->fail(RGBDR_ERR_STATE, "integrate with bricks before update_occupied_bricks");
  HIPCHK(hipSetDevice(ctx->device));
  const rgbdr_geometry& g = ctx->geo;
  IntegrateParams p{};
  p.N = N;
  p.W = ctx->cfg.depth_w;
  p.H = ctx->cfg.depth_h;
  p.X = g.res_volume[0];
  p.Y = g.res_volume[1];
  p.Z = g.res_volume[2];
  p.TX = g.tiles[0];
  p.TY = g.tiles[1];
  p.tz0 = g.slab_tile_z0;
  p.ntz = g.slab_tile_z1 - g.slab_tile_z0;
  p.limit = ctx->cfg.tsdf_limit;
  p.stepX = 1.0f / (float)p.X;
  p.stepY = 1.0f / (float)p.Y;
  p.stepZ = 1.0f / (float)p.Z;
  const size_t img = (size_t)p.W * p.H;
  for (int i = 0; i < N; ++i) {
    p.frame[i] = ctx->frame_buf(ctx->rbuf) + img * i;
    p.lut[i] = ctx->d_lut_generic[i];
    p.rx[i] = (int)ctx->inv_res[i][0];
    p.ry[i] = (int)ctx->inv_res[i][1];
    p.rz[i] = (int)ctx->inv_res[i][2];
    p.zoff[i] = ctx->zoff[i];
  }
  p.lut_tiled = ctx->d_lut_tiled;
  p.win = ctx->d_win;
  p.use_bricks = bricks ? 1 : 0;
  p.brick_mask = ctx->mask_buf(ctx->rbuf);
  p.brick_counters = nullptr;
  if (bricks && ctx->occ_lazy) {
    if (all_tiled) {  // k_brick_clear<true> filters on the way
      p.brick_counters = ctx->d_counters + (size_t)ctx->occ_lazy_cbuf * g.num_bricks;  // the buffer the update saw
      p.min_voxels = ctx->occ_lazy_min;
      p.brick_mask_out = ctx->mask_buf(ctx->rbuf);
      p.num_bricks = g.num_bricks;
      ctx->occ_lazy = false;
    } else {
      int rc_ = materialise_mask(ctx);
      if (rc_ != RGBDR_OK) return rc_;
    }
  }
  p.vbx = ctx->d_brick_tab;
  p.vby = p.vbx + g.res_volume[0];
  p.vbz = p.vby + g.res_volume[1];
  p.tbx = p.vbz + g.res_volume[2];
  p.tby = p.tbx + g.tiles[0];
  p.tbz = p.tby + g.tiles[1];
  p.twx = p.tbz + g.tiles[2];
  p.twy = p.twx + g.tiles[0];
  p.twz = p.twy + g.tiles[1];
  p.ovx = ctx->bt.overflow[0];
  p.ovy = ctx->bt.overflow[1];
  p.bx = g.res_bricks[0];
  p.by = g.res_bricks[1];
  p.bz = g.res_bricks[2];
  p.tsdf = ctx->d_tsdf_owned;
  p.tile_list = ctx->d_tile_list;
  p.tile_count = ctx->d_tile_list + (size_t)p.TX * p.TY * p.ntz + ctx->tile_count_parity;
  p.tile_count_next = ctx->d_tile_list + (size_t)p.TX * p.TY * p.ntz + (1 - ctx->tile_count_parity);
  p.tile_state = ctx->d_tile_state;
  const bool elide = !bricks && all_tiled && (ctx->cfg.flags & RGBDR_FLAG_ELIDE_STORES) != 0;
  p.elide_stores = elide ? 1 : 0;
  p.launches = (unsigned)ctx->sweep_launches;
  const bool skip_bg = !bricks && all_tiled && (ctx->cfg.flags & RGBDR_FLAG_SKIP_BACKGROUND) != 0 && p.limit > 0.0f;
  if (bricks && all_tiled) ctx->tile_count_parity ^= 1;
  p.skip_background = skip_bg ? 1 : 0;
  p.win_dmin = reinterpret_cast<const float*>(ctx->d_win + (size_t)p.TX * p.TY * p.ntz * N);
  p.win_dmax = reinterpret_cast<const float*>(ctx->d_win + 2 * (size_t)p.TX * p.TY * p.ntz * N);
  if ((!bricks && !elide && !skip_bg) || !all_tiled) {  // sweeps that overwrite tiles without keeping tile_state
    int rc_ = bump_clear_epoch(ctx);
    if (rc_ != RGBDR_OK) return rc_;
  }
  p.epoch = ctx->clear_epoch;
  const int sb = ctx->stage_target;
  const size_t face_floats = ctx->layer_floats * (size_t)ctx->halo;
  bool copy_faces = false;
  if (sb >= 0 && ctx->halo > 0) {
    float* lo = ctx->cfg.slab_rank > 0 ? ctx->d_stage[sb][0] : nullptr;
    float* hi = ctx->cfg.slab_rank < ctx->cfg.slab_count - 1 ? ctx->d_stage[sb][1] : nullptr;
    p.stage_layers = ctx->halo;
    if (integrate_stages_halo(p, all_tiled)) {
      p.stage_lo = lo;
      p.stage_hi = hi;
    } else {
      copy_faces = true;
    }
  }
  if (ctx->pipelined() && ctx->ev_pre_rec[ctx->rbuf]) {
    HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_pre[ctx->rbuf], 0));
    ctx->pre_joined = ctx->pre_serial;
  }
  { int rc_ = join_async_gather(ctx, ctx->stream); if (rc_ != RGBDR_OK) return rc_; }
  tbegin(ctx, "2integrate", ctx->stream);
  if (skip_bg) {
    int rc_ = skip_sweep(ctx, p);
    if (rc_ != RGBDR_OK) return rc_;
  } else {
    launch_integrate(p, all_tiled, ctx->stream);
  }
  tend(ctx, "2integrate", ctx->stream);
  LAUNCHCHK("integrate");
  if (copy_faces) {  // sweeps that do not stage by themselves: copy the boundary layers after them
    const int owned = g.slab_tile_z1 - g.slab_tile_z0;
    if (ctx->cfg.slab_rank > 0)
      HIPCHK(hipMemcpyAsync(ctx->d_stage[sb][0], ctx->d_tsdf_owned, face_floats * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
    if (ctx->cfg.slab_rank < ctx->cfg.slab_count - 1)
      HIPCHK(hipMemcpyAsync(ctx->d_stage[sb][1], ctx->d_tsdf_owned + ctx->layer_floats * (size_t)(owned - ctx->halo),
                            face_floats * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
  }
  ctx->integrated = true;
  ctx->tile_states_kept = all_tiled && (bricks || elide || skip_bg);
  if (sb >= 0 && ctx->halo > 0) ctx->halo_staged = true;  // the staging set of this step holds this sweep's faces
  if (ctx->pipelined()) {
    if (ctx->draw_expected) {  // (rgbdr_draw's record will cover this sweep: context.hpp)
      ctx->int_unrecorded[ctx->rbuf] = true;
    } else {
      HIPCHK(hipEventRecord(ctx->ev_int[ctx->rbuf], ctx->stream));
      ctx->ev_int_rec[ctx->rbuf] = true;
      ctx->int_unrecorded[ctx->rbuf] = false;
    }
    ctx->wbuf = ctx->rbuf ^ 1;  // the next frame's pre_* chain may run while this sweep reads rbuf
  }
  return RGBDR_OK;
}
RGBDR_CONTAIN(ctx)

int rgbdr_step(rgbdr_ctx* ctx, const void* depth, const void* color)
try {
  int rc = rgbdr_upload_frame(ctx, depth, color);
  if (rc == RGBDR_OK) rc = rgbdr_clear_occupied_bricks(ctx);
  if (rc == RGBDR_OK) rc = rgbdr_process_textures(ctx);
  if (rc == RGBDR_OK) rc = rgbdr_update_occupied_bricks(ctx);
  if (rc == RGBDR_OK) rc = rgbdr_integrate(ctx);
  return rc;
}
RGBDR_CONTAIN(ctx)

int rgbdr_sync(rgbdr_ctx* ctx)
try {
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  return sync_all(ctx);
}
RGBDR_CONTAIN(ctx)

// ---------------------------------------------------------------------------
int rgbdr_set_voxel_size(rgbdr_ctx* ctx, float size)
try {
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (!(size > 0.0f)) return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, "voxel size must be > 0");
  { int rc_ = sync_all(ctx); if (rc_ != RGBDR_OK) return rc_; }
  rgbdr_config old = ctx->cfg;
  ctx->cfg.voxel_size = size;
  ctx->cfg.res_override[0] = ctx->cfg.res_override[1] = ctx->cfg.res_override[2] = 0;
  rgbdr_geometry g;
  std::string e;
  int rc = compute_geometry(ctx->cfg, &g, &e);
  if (rc != RGBDR_OK) {
    ctx->cfg = old;
    return ctx->fail(rc, e);
  }
  return alloc_volume(ctx);
}
RGBDR_CONTAIN(ctx)

int rgbdr_set_brick_size(rgbdr_ctx* ctx, float size)
try {
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (!(size > 0.0f)) return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, "brick size must be > 0");
  { int rc_ = sync_all(ctx); if (rc_ != RGBDR_OK) return rc_; }
  rgbdr_config trial = ctx->cfg;
  trial.brick_size = size;
  rgbdr_geometry g;
  std::string e;
  int rc = compute_geometry(trial, &g, &e);
  if (rc != RGBDR_OK) return ctx->fail(rc, e);
  // only the brick table changes; the volume and the LUTs stay
  rc = alloc_brick_table(ctx, trial, g);
  if (rc != RGBDR_OK) return rc;  // the old table is still in place
  ctx->cfg = trial;
  ctx->geo = g;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return RGBDR_OK;
}
RGBDR_CONTAIN(ctx)

int rgbdr_set_tsdf_limit(rgbdr_ctx* ctx, float limit)
try {
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (!(limit >= kMinTsdfLimit) || !std::isfinite(limit))
    return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, "tsdf limit must be a finite number >= 1e-6 (the ray-marcher steps limit / 2 through the unit cube)");
  ctx->cfg.tsdf_limit = limit;
  return bump_clear_epoch(ctx);  // tiles cleared to the old -limit no longer count as cleared
}
RGBDR_CONTAIN(ctx)

static int set_flag(rgbdr_ctx* ctx, uint32_t flag, int on)
{
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (on)
    ctx->cfg.flags |= flag;
  else
    ctx->cfg.flags &= ~flag;
  return RGBDR_OK;
}
int rgbdr_set_use_bricks(rgbdr_ctx* ctx, int on) { return set_flag(ctx, RGBDR_FLAG_USE_BRICKS, on); }
int rgbdr_set_pipelined(rgbdr_ctx* ctx, int on)
try {
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  int rc = materialise_mask(ctx);  // (the lazy filter is a single-stream shortcut)
  if (rc == RGBDR_OK) rc = sync_all(ctx);
  if (rc != RGBDR_OK) return rc;
  ctx->wbuf = ctx->rbuf;  // keep reading what was written last
  ctx->ev_pre_rec[0] = ctx->ev_pre_rec[1] = ctx->ev_int_rec[0] = ctx->ev_int_rec[1] = false;
  ctx->int_unrecorded[0] = ctx->int_unrecorded[1] = ctx->draw_expected = false;  // (everything has completed)
  ctx->ev_view_rec[0] = ctx->ev_view_rec[1] = false;
  return set_flag(ctx, RGBDR_FLAG_PIPELINE, on);
}
RGBDR_CONTAIN(ctx)
int rgbdr_set_elide_stores(rgbdr_ctx* ctx, int on) { return set_flag(ctx, RGBDR_FLAG_ELIDE_STORES, on); }
int rgbdr_set_skip_background(rgbdr_ctx* ctx, int on) { return set_flag(ctx, RGBDR_FLAG_SKIP_BACKGROUND, on); }
int rgbdr_set_sweep_launches(rgbdr_ctx* ctx, int n)
try {
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (n < 1 || n > 64) return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, "set_sweep_launches: 1 .. 64");
  ctx->sweep_launches = n;
  return RGBDR_OK;
}
RGBDR_CONTAIN(ctx)
int rgbdr_filter_textures(rgbdr_ctx* ctx, int on) { return set_flag(ctx, RGBDR_FLAG_FILTER, on); }
int rgbdr_use_processed_depths(rgbdr_ctx* ctx, int on) { return set_flag(ctx, RGBDR_FLAG_PROCESSED, on); }
int rgbdr_refine_boundary(rgbdr_ctx* ctx, int on) { return set_flag(ctx, RGBDR_FLAG_REFINE, on); }

int rgbdr_set_min_voxels_per_brick(rgbdr_ctx* ctx, uint32_t n)
try {
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  ctx->cfg.min_voxels_per_brick = n;
  return RGBDR_OK;
}
RGBDR_CONTAIN(ctx)

float rgbdr_get_brick_size(const rgbdr_ctx* ctx) { return ctx ? ctx->geo.brick_size : 0.0f; }
uint32_t rgbdr_num_bricks(const rgbdr_ctx* ctx) { return ctx ? (uint32_t)ctx->geo.num_bricks : 0u; }

float rgbdr_occupied_ratio(rgbdr_ctx* ctx)
{
  if (!ctx || !ctx->mask_valid) return 0.0f;
  size_t n = 0;
  float ratio = 0.0f;
  if (rgbdr_get_occupied(ctx, nullptr, 0, &n, &ratio) != RGBDR_OK) return 0.0f;
  return ratio;
}

int rgbdr_get_geometry(const rgbdr_ctx* ctx, rgbdr_geometry* out)
try {
  if (!ctx || !out) return RGBDR_ERR_INVALID_ARGUMENT;
  *out = ctx->geo;
  return RGBDR_OK;
}
RGBDR_CONTAIN(nullptr)

int rgbdr_get_camera_position(const rgbdr_ctx* ctx, int sensor, float out[3])
try {
  if (!ctx || !out) return RGBDR_ERR_INVALID_ARGUMENT;
  if (sensor < 0 || sensor >= ctx->cfg.num_sensors || !ctx->have_calib[sensor]) return RGBDR_ERR_OUT_OF_RANGE;
  std::memcpy(out, ctx->cam_pos[sensor], 12);
  return RGBDR_OK;
}
RGBDR_CONTAIN(nullptr)

// ---------------------------------------------------------------------------
int rgbdr_readback_tsdf(rgbdr_ctx* ctx, float* dst)
try {
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (!dst) return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, "null destination");
  { int rc_ = sync_all(ctx); if (rc_ != RGBDR_OK) return rc_; }
  HIPCHK(hipSetDevice(ctx->device));
  const rgbdr_geometry& g = ctx->geo;
  const size_t n = (size_t)g.res_volume[0] * g.res_volume[1] * (size_t)(g.slab_voxel_z1 - g.slab_voxel_z0);
  if (ctx->linear_floats < n) {
    (void)hipFree(ctx->d_linear);
    ctx->d_linear = nullptr;
    ctx->linear_floats = 0;
    HIPCHK(hipMalloc((void**)&ctx->d_linear, n * sizeof(float)));
    ctx->linear_floats = n;
  }
  launch_detile(ctx->d_tsdf_owned, ctx->d_linear, g.res_volume[0], g.res_volume[1], g.tiles[0], g.tiles[1],
                g.slab_tile_z0, g.slab_voxel_z0, g.slab_voxel_z1, ctx->stream);
  LAUNCHCHK("detile");
  HIPCHK(hipMemcpyAsync(dst, ctx->d_linear, n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return RGBDR_OK;
}
RGBDR_CONTAIN(ctx)

int rgbdr_readback_image(rgbdr_ctx* ctx, int which, int sensor, float* dst)
try {
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (!dst) return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, "null destination");
  if (sensor < 0 || sensor >= nsens(ctx)) return ctx->fail(RGBDR_ERR_OUT_OF_RANGE, "sensor index out of range");
  const float* src = nullptr;
  int ch = 1;
  switch (which) {
    case RGBDR_IMG_DEPTH_RAW: src = ctx->d_depth_raw; break;
    case RGBDR_IMG_DEPTH_MORPH: src = ctx->d_depth_morph; break;
    case RGBDR_IMG_DEPTH_RG: src = ctx->d_depth_rg; ch = 2; break;
    case RGBDR_IMG_LAB: src = ctx->d_lab; ch = 3; break;
    case RGBDR_IMG_DEPTH_B_RG: src = ctx->d_depth_b; ch = 2; break;
    case RGBDR_IMG_SILHOUETTE: src = ctx->d_sil; break;
    case RGBDR_IMG_NORMAL: src = ctx->d_normal; ch = 3; break;
    case RGBDR_IMG_QUALITY: src = ctx->d_quality; break;
    default: return ctx->fail(RGBDR_ERR_OUT_OF_RANGE, "unknown image id");
  }
  { int rc_ = sync_all(ctx); if (rc_ != RGBDR_OK) return rc_; }
  HIPCHK(hipSetDevice(ctx->device));
  const size_t img = (size_t)ctx->cfg.depth_w * ctx->cfg.depth_h * ch;
  HIPCHK(hipMemcpyAsync(dst, src + img * sensor, img * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return RGBDR_OK;
}
RGBDR_CONTAIN(ctx)

int rgbdr_device_image(rgbdr_ctx* ctx, int which, int sensor, rgbdr_image_device_view* out)
try {
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (!out) return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, "null view");
  if (sensor < 0 || sensor >= nsens(ctx)) return ctx->fail(RGBDR_ERR_OUT_OF_RANGE, "sensor index out of range");
  const size_t px = (size_t)ctx->cfg.depth_w * ctx->cfg.depth_h;
  rgbdr_image_device_view v{};
  v.width = ctx->cfg.depth_w;
  v.height = ctx->cfg.depth_h;
  v.channels = 1;
  v.element_bytes = 4;
  const float* f = nullptr;
  switch (which) {
    case RGBDR_IMG_DEPTH_RAW: f = ctx->d_depth_raw; break;
    case RGBDR_IMG_DEPTH_MORPH: f = ctx->d_depth_morph; break;
    case RGBDR_IMG_DEPTH_RG: f = ctx->d_depth_rg; v.channels = 2; break;
    case RGBDR_IMG_LAB: f = ctx->d_lab; v.channels = 3; break;
    case RGBDR_IMG_DEPTH_B_RG: f = ctx->d_depth_b; v.channels = 2; break;
    case RGBDR_IMG_SILHOUETTE: f = ctx->d_sil; break;
    case RGBDR_IMG_NORMAL: f = ctx->d_normal; v.channels = 3; break;
    case RGBDR_IMG_QUALITY: f = ctx->d_quality; break;
    case RGBDR_IMG_COLOR:
      { int rc_ = ensure_color_decoded(ctx); if (rc_ != RGBDR_OK) return rc_; }
      ctx->color_view_out = true;  // from now on every upload of DXT frames decodes them as well
      v.width = ctx->cfg.color_w;
      v.height = ctx->cfg.color_h;
      v.channels = 3;
      v.element_bytes = 1;
      v.ptr = ctx->color_half(ctx->color_up) + (size_t)ctx->cfg.color_w * ctx->cfg.color_h * 3 * sensor;  // (uploads stay in this half from now on)
      break;
    default: return ctx->fail(RGBDR_ERR_OUT_OF_RANGE, "unknown image id");
  }
  if (f) v.ptr = (void*)(f + px * v.channels * sensor);
  v.stream = ctx->pstream();
  *out = v;
  return RGBDR_OK;
}
RGBDR_CONTAIN(ctx)

int rgbdr_readback_color(rgbdr_ctx* ctx, int sensor, uint8_t* dst)
try {
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (!dst) return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, "null destination");
  if (sensor < 0 || sensor >= nsens(ctx)) return ctx->fail(RGBDR_ERR_OUT_OF_RANGE, "sensor index out of range");
  { int rc_ = ensure_color_decoded(ctx); if (rc_ != RGBDR_OK) return rc_; }
  { int rc_ = sync_all(ctx); if (rc_ != RGBDR_OK) return rc_; }
  const size_t img = (size_t)ctx->cfg.color_w * ctx->cfg.color_h * 3;
  HIPCHK(hipMemcpyAsync(dst, ctx->color_half(ctx->color_up) + img * sensor, img, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return RGBDR_OK;
}
RGBDR_CONTAIN(ctx)

int rgbdr_readback_brick_counters(rgbdr_ctx* ctx, uint32_t* dst)
try {
  if (ctx) { int rc_ = flush_clear(ctx); if (rc_ != RGBDR_OK) return rc_; }
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (!dst) return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, "null destination");
  { int rc_ = sync_all(ctx); if (rc_ != RGBDR_OK) return rc_; }
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipMemcpyAsync(dst, ctx->counters_cur(), (size_t)ctx->geo.num_bricks * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return RGBDR_OK;
}
RGBDR_CONTAIN(ctx)

int rgbdr_get_occupied(rgbdr_ctx* ctx, uint32_t* ids, size_t capacity, size_t* count, float* ratio)
try {
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (!count) return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, "null count");
  if (!ctx->mask_valid) return ctx->fail(RGBDR_ERR_STATE, "get_occupied before update_occupied_bricks");
  { int rc_ = materialise_mask(ctx); if (rc_ != RGBDR_OK) return rc_; }
  { int rc_ = sync_all(ctx); if (rc_ != RGBDR_OK) return rc_; }
  HIPCHK(hipSetDevice(ctx->device));
  launch_compact_occupied(ctx->mask_buf(ctx->rbuf), (uint32_t)ctx->geo.num_bricks, ctx->d_ids,
                          ctx->count_buf(ctx->rbuf) + 1, ctx->stream);
  LAUNCHCHK("compact_occupied");
  uint32_t c = 0;
  HIPCHK(hipMemcpyAsync(&c, ctx->count_buf(ctx->rbuf) + 1, 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  *count = c;
  if (ratio) *ratio = (float)c / (float)ctx->geo.num_bricks;
  if (ids) {
    if (capacity < c) return ctx->fail(RGBDR_ERR_OUT_OF_RANGE, "id buffer too small");
    HIPCHK(hipMemcpyAsync(ids, ctx->d_ids, (size_t)c * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
  }
  return RGBDR_OK;
}
RGBDR_CONTAIN(ctx)

int rgbdr_device_tsdf(rgbdr_ctx* ctx, rgbdr_tsdf_device_view* out)
try {
  if (!ctx || !out) return RGBDR_ERR_INVALID_ARGUMENT;
  out->base = ctx->d_tsdf_base;
  out->owned = ctx->d_tsdf_owned;
  out->layer_bytes = ctx->layer_floats * sizeof(float);
  out->owned_layers = ctx->geo.slab_tile_z1 - ctx->geo.slab_tile_z0;
  out->halo_layers = ctx->halo;
  return RGBDR_OK;
}
RGBDR_CONTAIN(ctx)

int rgbdr_device_frame(rgbdr_ctx* ctx, int sensor, void** ptr)
try {
  if (!ctx || !ptr) return RGBDR_ERR_INVALID_ARGUMENT;
  if (sensor < 0 || sensor >= nsens(ctx)) return ctx->fail(RGBDR_ERR_OUT_OF_RANGE, "sensor index out of range");
  *ptr = ctx->frame_buf(ctx->rbuf) + (size_t)ctx->cfg.depth_w * ctx->cfg.depth_h * sensor;
  return RGBDR_OK;
}
RGBDR_CONTAIN(ctx)

void* rgbdr_stream(rgbdr_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int rgbdr_set_stream(rgbdr_ctx* ctx, void* hip_stream)
try {
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  { int rc_ = sync_all(ctx); if (rc_ != RGBDR_OK) return rc_; }
  ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
  return RGBDR_OK;
}
RGBDR_CONTAIN(ctx)

}  // extern "C"
