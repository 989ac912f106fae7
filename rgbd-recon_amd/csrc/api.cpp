// api.cpp -- the C ABI of include/rgbdr.h: context, device memory, call order.
// Host-side mirror of NetKinectArray / CalibVolumes / ReconIntegration state;
// every GL texture unit / SSBO binding of the reference is a
// device pointer owned by the context here.  There is no CPU fallback: without a
// HIP device rgbdr_create fails with RGBDR_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "rgbdr_internal.hpp"

using namespace rgbdr;

namespace {
thread_local std::string g_create_error;

// One named interval.  In accumulate mode every begin/end takes a fresh event
// pair so a whole timed region can be resolved afterwards without a host sync
// inside it (bench.py reads the kernel's average launch duration that way).
struct Timer {
  hipEvent_t a = nullptr, b = nullptr;
  bool recorded = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending, pool;
};
}  // namespace

struct rgbdr_ctx {
  rgbdr_config cfg{};
  rgbdr_geometry geo{};
  int device = 0;
  hipStream_t stream = nullptr;      // where work is enqueued
  hipStream_t own_stream = nullptr;  // created with the context
  // Pipelined mode (RGBDR_FLAG_PIPELINE): upload + pre_* chain + occupied update of
  // frame k+1 run on pre_stream while integrate of frame k runs on `stream`.  The
  // only state both touch -- packed frame, occupied mask, occupied count -- is
  // double buffered; events order producer and consumer of each buffer.
  hipStream_t pre_stream = nullptr;
  int wbuf = 0, rbuf = 0;            // buffer the next process_textures writes / the latest one written
  hipEvent_t ev_pre[2] = {nullptr, nullptr}, ev_int[2] = {nullptr, nullptr};
  bool ev_pre_rec[2] = {false, false}, ev_int_rec[2] = {false, false};
  bool pipelined() const { return (cfg.flags & RGBDR_FLAG_PIPELINE) != 0; }
  hipStream_t pstream() const { return pipelined() ? pre_stream : stream; }
  uint2* frame_buf(int b) const { return d_frame + (size_t)b * cfg.num_sensors * cfg.depth_w * cfg.depth_h; }
  uint8_t* mask_buf(int b) const { return d_mask + (size_t)b * geo.num_bricks; }
  uint32_t* count_buf(int b) const { return d_count + 4 * b; }
  std::string err;

  // images ([N][H][W][c])
  float *d_depth_raw = nullptr, *d_depth_morph = nullptr, *d_depth_rg = nullptr, *d_lab = nullptr;
  float *d_depth_b = nullptr, *d_sil = nullptr, *d_normal = nullptr, *d_quality = nullptr;
  uint2* d_frame = nullptr;
  uint8_t *d_color = nullptr, *d_depth_u8 = nullptr, *d_color_dxt = nullptr;
  bool frame_uploaded = false, textures_processed = false;

  // forward calibration
  float4* d_cv_xyz[kMaxSensors] = {};
  float2* d_cv_uv[kMaxSensors] = {};
  uint32_t xyz_res[kMaxSensors][3] = {}, uv_res[kMaxSensors][3] = {};
  float min_ds[kMaxSensors] = {}, max_ds[kMaxSensors] = {};
  float cam_pos[kMaxSensors][3] = {};
  float planes[kMaxSensors][6][4] = {};  // Frustum::getPlanes of cv_xyz
  bool have_calib[kMaxSensors] = {};

  // inverse calibration
  bool inv_set[kMaxSensors] = {};
  bool inv_tiled[kMaxSensors] = {};
  bool inv_resampled[kMaxSensors] = {};  // tiled planes hold the LUT resampled at voxel centres
  uint32_t inv_res[kMaxSensors][3] = {};
  float* d_lut_tiled = nullptr;       // grid-layout LUT planes of the OWNED tile layers ...
  float* d_lut_tiled_base = nullptr;  // ... inside an allocation with `halo` more layers on each side
  // double_pbo of NetKinectArray (double_pixel_buffer.cpp:35-81): two page-locked host frame
  // sets; the producer fills the back one, upload_mapped swaps and DMAs from the front one
  void* h_depth[2] = {nullptr, nullptr};
  void* h_color[2] = {nullptr, nullptr};
  hipEvent_t ev_mapped[2] = {nullptr, nullptr};
  bool ev_mapped_rec[2] = {false, false};
  int mapped_back = 0;
  int32_t* d_win = nullptr;  // per (tile, sensor) frame-window origin
  float arena_probe_ms[16] = {0};  // LUT-stream time of each candidate placement of the arena
  int arena_trials = 0, arena_chosen = 0;
  float4* d_lut_generic[kMaxSensors] = {};
  int zoff[kMaxSensors] = {};

  // volume
  float *d_tsdf_base = nullptr, *d_tsdf_owned = nullptr;
  size_t layer_floats = 0;
  int halo = 0;
  float* d_linear = nullptr;  // readback scratch
  size_t linear_floats = 0;
  float* d_view = nullptr;    // ray-march outputs: rgba, depth, samples
  size_t view_pixels = 0;
  int view_w = 0, view_h = 0; // size of the last ray-marched frame
  float* d_fill = nullptr;    // hole-fill atlases (2 x colour + depth) and the filled frame
  size_t fill_floats = 0;
  bool integrated = false;

  // bricks
  uint32_t *d_counters = nullptr, *d_ids = nullptr, *d_count = nullptr;
  bool clear_pending = false;       // clearOccupiedBricks was called; the zeroing rides on the next k_morph
  uint32_t* d_tile_list = nullptr;  // brick-skipping sweep: work list of owned tiles + its length (last entry)
  uint32_t* d_tile_state = nullptr; // per owned tile: epoch of the brick sweep since which it holds -limit (0: never)
  int tile_count_parity = 0;        // which of the two list counters the next brick sweep appends to
  // halo staging for Z slabs: two sets of (lower face, upper face) buffers of `halo` tile layers
  float* d_stage[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
  int stage_target = -1;            // set the next integrate fills (-1: none)
  uint32_t clear_epoch = 1;         // bumped whenever the volume may have been written by anything else
  uint8_t* d_mask = nullptr;
  bool mask_valid = false;
  // brick -> voxel membership of divideBox / containedVoxels (geometry.cpp compute_brick_tables):
  // device copy of vox[x] | vox[y] | vox[z] | tile[x] | tile[y] | tile[z]
  BrickTables bt;
  uint32_t* d_brick_tab = nullptr;

  bool timers = false, accumulate = false;
  int timer_detail = 2;  // 1: only "1preprocess" / "2integrate" / "bricks" ...; 2: also the five pre_* passes
  std::map<std::string, Timer> tm;

  int fail(int code, const std::string& m)
  {
    err = m;
    return code;
  }
};

#define HIPCHK(expr)                                                                                       \
  do {                                                                                                     \
    hipError_t e_ = (expr);                                                                                \
    if (e_ != hipSuccess)                                                                                  \
      return ctx->fail(RGBDR_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));                  \
  } while (0)

#define LAUNCHCHK(what)                                                                                    \
  do {                                                                                                     \
    hipError_t e_ = hipGetLastError();                                                                     \
    if (e_ != hipSuccess) return ctx->fail(RGBDR_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e_)); \
  } while (0)

// device scratch that is released on every return path (HIPCHK / LAUNCHCHK return early)
struct DevScratch {
  void* p = nullptr;
  ~DevScratch() { (void)hipFree(p); }
  template <class T> T* as() const { return (T*)p; }
};

// Whatever writes the volume without keeping tile_state (full sweep, generic-LUT sweep, stream
// replay) or changes -limit invalidates every recorded "this tile already holds -limit".
static int bump_clear_epoch(rgbdr_ctx* ctx)
{
  if (++ctx->clear_epoch == 0) {  // wrapped: forget every recorded clear
    const rgbdr_geometry& g = ctx->geo;
    const size_t n = (size_t)g.tiles[0] * g.tiles[1] * (size_t)(g.slab_tile_z1 - g.slab_tile_z0);
    if (ctx->d_tile_state) HIPCHK(hipMemsetAsync(ctx->d_tile_state, 0, n * sizeof(uint32_t), ctx->stream));
    ctx->clear_epoch = 1;
  }
  return RGBDR_OK;
}

static int nsens(const rgbdr_ctx* c) { return c->cfg.num_sensors; }
// bytes of one sensor's colour frame as the caller hands it over (NetKinectArray.cpp:120-131)
static size_t color_frame_bytes(const rgbdr_config& c)
{
  const size_t blocks = (size_t)((c.color_w + 3) / 4) * ((c.color_h + 3) / 4);
  if (c.compress_rgb == 1) return blocks * 8;
  if (c.compress_rgb == 5) return blocks * 16;
  return (size_t)c.color_w * c.color_h * 3;
}
static size_t npx(const rgbdr_ctx* c) { return (size_t)c->cfg.num_sensors * c->cfg.depth_w * c->cfg.depth_h; }

// the per-pass timers sit inside "1preprocess"; every timer is two event records on
// the stream, so a host that only wants the totals can switch them off (detail 1)
static bool timer_is_pass(const char* n) { return n[0] == 'm' || (n[0] == 'b' && n[1] != 'r') || n[0] == 'n' || n[0] == 'q'; }
// detail 2: every timer; 1: the totals; 0: "2integrate" alone (an event record costs ~4 us of stream time)
static bool timer_muted(const rgbdr_ctx* c, const char* n)
{
  if (!c->timers) return true;
  if (c->timer_detail < 1) return !(n[0] == '2');
  return c->timer_detail < 2 && timer_is_pass(n);
}

static void tbegin(rgbdr_ctx* c, const char* name, hipStream_t st)
{
  if (timer_muted(c, name)) return;
  Timer& t = c->tm[name];
  if (c->accumulate) {
    std::pair<hipEvent_t, hipEvent_t> ev{nullptr, nullptr};
    if (!t.pool.empty()) {
      ev = t.pool.back();
      t.pool.pop_back();
    } else {
      (void)hipEventCreate(&ev.first);
      (void)hipEventCreate(&ev.second);
    }
    t.pending.push_back(ev);
    (void)hipEventRecord(ev.first, st);
    return;
  }
  if (!t.a) {
    (void)hipEventCreate(&t.a);
    (void)hipEventCreate(&t.b);
  }
  (void)hipEventRecord(t.a, st);
}
static void tend(rgbdr_ctx* c, const char* name, hipStream_t st)
{
  if (timer_muted(c, name)) return;
  Timer& t = c->tm[name];
  if (c->accumulate) {
    if (!t.pending.empty()) (void)hipEventRecord(t.pending.back().second, st);
    return;
  }
  (void)hipEventRecord(t.b, st);
  t.recorded = true;
}

static void free_volume(rgbdr_ctx* c)
{
  (void)hipFree(c->d_tsdf_base);
  (void)hipFree(c->d_linear);
  (void)hipFree(c->d_view);
  (void)hipFree(c->d_fill);
  c->d_view = c->d_fill = nullptr;
  c->view_pixels = c->fill_floats = 0;
  c->view_w = c->view_h = 0;
  c->integrated = false;
  for (int b = 0; b < 2; ++b)
    for (int f = 0; f < 2; ++f) {
      (void)hipFree(c->d_stage[b][f]);
      c->d_stage[b][f] = nullptr;
    }
  c->stage_target = -1;
  (void)hipFree(c->d_tile_list);
  (void)hipFree(c->d_tile_state);
  c->d_tile_list = c->d_tile_state = nullptr;
  (void)hipFree(c->d_counters);
  (void)hipFree(c->d_ids);
  (void)hipFree(c->d_mask);
  (void)hipFree(c->d_brick_tab);
  c->d_brick_tab = nullptr;
  (void)hipFree(c->d_lut_tiled_base);
  c->d_lut_tiled_base = nullptr;
  (void)hipFree(c->d_win);
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
  uint32_t *counters = nullptr, *ids = nullptr, *tab = nullptr;
  uint8_t* mask = nullptr;
  const size_t nb = (size_t)g.num_bricks;
  if (hipMalloc((void**)&counters, nb * sizeof(uint32_t)) != hipSuccess || hipMalloc((void**)&ids, nb * sizeof(uint32_t)) != hipSuccess ||
      hipMalloc((void**)&mask, nb * 2) != hipSuccess || hipMalloc((void**)&tab, host.size() * sizeof(uint32_t)) != hipSuccess ||
      hipMemcpy(tab, host.data(), host.size() * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemsetAsync(counters, 0, nb * sizeof(uint32_t), ctx->stream) != hipSuccess ||
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
  ctx->d_ids = ids;
  ctx->d_mask = mask;
  ctx->d_brick_tab = tab;
  ctx->bt = std::move(bt);
  ctx->mask_valid = false;
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

// drain both streams (readbacks, setters, resizes)
static int sync_all(rgbdr_ctx* ctx)
{
  HIPCHK(hipSetDevice(ctx->device));
  if (ctx->pre_stream) HIPCHK(hipStreamSynchronize(ctx->pre_stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return RGBDR_OK;
}

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
    default: return "unknown status";
  }
}

const char* rgbdr_last_error(const rgbdr_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int rgbdr_compute_geometry(const rgbdr_config* cfg, rgbdr_geometry* out)
{
  if (!cfg || !out) return RGBDR_ERR_INVALID_ARGUMENT;
  return compute_geometry(*cfg, out, &g_create_error);
}

int rgbdr_brick_voxel_range(const rgbdr_config* cfg, int axis, int brick, int32_t* first, int32_t* last)
{
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

int rgbdr_slab_range(int tiles_z, int count, int rank, int* t0, int* t1)
{
  if (!t0 || !t1) return RGBDR_ERR_INVALID_ARGUMENT;
  return slab_range(tiles_z, count, rank, t0, t1);
}

int rgbdr_camera_position(const rgbdr_lut* cv_xyz, float out[3])
{
  if (!cv_xyz || !cv_xyz->data || !out) return RGBDR_ERR_INVALID_ARGUMENT;
  if (cv_xyz->res[0] < 1 || cv_xyz->res[1] < 1 || cv_xyz->res[2] < 1) return RGBDR_ERR_INVALID_ARGUMENT;
  camera_position((const float*)cv_xyz->data, cv_xyz->res, out);
  return RGBDR_OK;
}

int rgbdr_create(const rgbdr_config* cfg, int device_id, rgbdr_ctx** out)
{
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
  if (!(cfg->tsdf_limit > 0.0f)) return bad(RGBDR_ERR_INVALID_ARGUMENT, "tsdf_limit must be > 0");
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
  ctx->cfg = *cfg;
  if (ctx->cfg.slab_count <= 0) {
    ctx->cfg.slab_count = 1;
    ctx->cfg.slab_rank = 0;
  }
  ctx->device = device_id;
  auto cleanup = [&](int code) {
    g_create_error = ctx->err;
    rgbdr_destroy(ctx);
    return code;
  };
  if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) {
    ctx->err = "hipStreamCreate failed";
    return cleanup(RGBDR_ERR_HIP);
  }
  ctx->stream = ctx->own_stream;
  {
    // the small pre_* kernels of the next frame must not queue behind the 262 144 workgroups
    // of an integrate sweep: give their stream the highest dispatch priority
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    if (hipStreamCreateWithPriority(&ctx->pre_stream, hipStreamNonBlocking, greatest) != hipSuccess) {
      ctx->err = "hipStreamCreate failed";
      return cleanup(RGBDR_ERR_HIP);
    }
  }
  for (int b = 0; b < 2; ++b)
    if (hipEventCreateWithFlags(&ctx->ev_pre[b], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_int[b], hipEventDisableTiming) != hipSuccess) {
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
                {(void**)&ctx->d_frame, n * 8 * 2},      {(void**)&ctx->d_color, ncol},
                {(void**)&ctx->d_depth_u8, n},        {(void**)&ctx->d_count, 32},
                {(void**)&ctx->d_color_dxt, color_frame_bytes(*cfg) * cfg->num_sensors}};
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
  *out = ctx;
  return RGBDR_OK;
}

void rgbdr_destroy(rgbdr_ctx* ctx)
{
  if (ctx) {
    for (int b = 0; b < 2; ++b) {
      if (ctx->h_depth[b]) (void)hipHostFree(ctx->h_depth[b]);
      if (ctx->h_color[b]) (void)hipHostFree(ctx->h_color[b]);
      if (ctx->ev_mapped[b]) (void)hipEventDestroy(ctx->ev_mapped[b]);
      ctx->h_depth[b] = ctx->h_color[b] = nullptr;
      ctx->ev_mapped[b] = nullptr;
    }
  }
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->pre_stream) (void)hipStreamSynchronize(ctx->pre_stream);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  free_volume(ctx);
  for (int b = 0; b < 2; ++b) {
    if (ctx->ev_pre[b]) (void)hipEventDestroy(ctx->ev_pre[b]);
    if (ctx->ev_int[b]) (void)hipEventDestroy(ctx->ev_int[b]);
  }
  if (ctx->pre_stream) (void)hipStreamDestroy(ctx->pre_stream);
  void* ptrs[] = {ctx->d_depth_raw, ctx->d_depth_morph, ctx->d_depth_rg, ctx->d_lab,   ctx->d_depth_b, ctx->d_sil,
                  ctx->d_normal,    ctx->d_quality,     ctx->d_frame,    ctx->d_color, ctx->d_depth_u8, ctx->d_count,
                  ctx->d_color_dxt};
  for (void* p : ptrs) (void)hipFree(p);
  for (int i = 0; i < kMaxSensors; ++i) {
    (void)hipFree(ctx->d_cv_xyz[i]);
    (void)hipFree(ctx->d_cv_uv[i]);
  }
  for (auto& kv : ctx->tm) {
    if (kv.second.a) (void)hipEventDestroy(kv.second.a);
    if (kv.second.b) (void)hipEventDestroy(kv.second.b);
    for (auto* v : {&kv.second.pending, &kv.second.pool})
      for (auto& ev : *v) {
        (void)hipEventDestroy(ev.first);
        (void)hipEventDestroy(ev.second);
      }
  }
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
  delete ctx;
}

// ---------------------------------------------------------------------------
int rgbdr_set_calibration(rgbdr_ctx* ctx, int sensor, const rgbdr_lut* xyz, const rgbdr_lut* uv)
{
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (sensor < 0 || sensor >= nsens(ctx)) return ctx->fail(RGBDR_ERR_OUT_OF_RANGE, "sensor index out of range");
  if (!xyz || !uv || !xyz->data || !uv->data) return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, "null calibration volume");
  for (int a = 0; a < 3; ++a)
    if (xyz->res[a] < 1 || uv->res[a] < 1) return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, "empty calibration volume");
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
  ctx->have_calib[sensor] = true;
  return RGBDR_OK;
}

// Tile layers [t0, t1) of the grid-layout LUT that are resident: the owned layers plus
// `halo` layers on each side where the volume has them; dst = where layer t0 lives.
struct LutExtent {
  int t0, t1, vz0, vz1;
  float* dst;
};
static LutExtent lut_extent(const rgbdr_ctx* ctx)
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

static int ensure_tiled_lut(rgbdr_ctx* ctx)
{
  if (ctx->d_lut_tiled) return RGBDR_OK;
  const rgbdr_geometry& g = ctx->geo;
  const size_t ntiles = (size_t)g.tiles[0] * g.tiles[1] * (g.slab_tile_z1 - g.slab_tile_z0);
  const size_t layer = (size_t)g.tiles[0] * g.tiles[1] * nsens(ctx) * 3 * kTileVoxels;
  const size_t layers = (size_t)(g.slab_tile_z1 - g.slab_tile_z0) + 2 * (size_t)ctx->halo;
  const size_t bytes = layer * layers * sizeof(float);
  // Where the driver places this arena shifts the sweep time of integrate by a few per cent on some
  // boxes (stable per allocation).  OPT-IN (RGBDR_ARENA_TRIALS=n, 2..16; default 1 = take
  // the first allocation, no probing): time the kernel's memory streams on up to n candidate
  // placements, keep the fastest.  Candidates are held while probing (otherwise the next hipMalloc
  // returns the same place), so this transiently needs up to n x the arena; it stops at the first
  // candidate at the fast level, when less than arena + 4 GiB is free, or after ~1 s.
  int trials = 1;
  if (const char* e = std::getenv("RGBDR_ARENA_TRIALS")) trials = std::atoi(e);
  if (trials > 16) trials = 16;
  if (trials < 1 || bytes < ((size_t)256 << 20)) trials = 1;  // small arenas: nothing to gain
  float* cand[16] = {nullptr};
  float* sink = ctx->d_tsdf_owned;  // the probe replays the TSDF store stream too: the volume is invalidated below
  float best_ms = 0.0f;
  int best = -1, got = 0;
  bool probed = false;
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int t = 0; t < trials; ++t) {
    size_t free_b = 0, total_b = 0;
    if (t > 0 && (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < bytes + ((size_t)4 << 30))) break;
    if (hipMalloc((void**)&cand[t], bytes) != hipSuccess) {
      (void)hipGetLastError();
      cand[t] = nullptr;
      if (t == 0) return ctx->fail(RGBDR_ERR_HIP, "hipMalloc of the inverse-LUT arena failed: out of device memory");
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
    if (ms > 0.0f && stream_bytes / (ms * 1e-3) >= 6.6e12) break;
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if ((double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec) > 1.0) break;
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
  // Releasing that much memory slows the device down for a moment (the driver wipes released VRAM
  // in the background): wait, at most 2 s, until the kept arena streams as it did when it was chosen.
  if (freed > 0 && best_ms > 0.0f) {
    for (int k = 0; k < 40; ++k) {
      const float ms = probe_arena_ms(cand[best] + layer * ctx->halo, ntiles, nsens(ctx), g.tiles[0], sink, ctx->stream);
      if (!(ms > best_ms * 1.01f)) break;
      struct timespec ts = {0, 50000000};
      nanosleep(&ts, nullptr);
    }
  }
  ctx->d_lut_tiled_base = cand[best];
  ctx->d_lut_tiled = ctx->d_lut_tiled_base + layer * ctx->halo;
  if (probed) {  // the replay stored into the volume: clear it again, forget recorded clears, nothing is integrated
    HIPCHK(hipMemsetAsync(ctx->d_tsdf_owned, 0, ntiles * kTileVoxels * sizeof(float), ctx->stream));
    { int rc_ = bump_clear_epoch(ctx); if (rc_ != RGBDR_OK) return rc_; }
    ctx->integrated = false;
  }
  HIPCHK(hipMalloc((void**)&ctx->d_win, ntiles * nsens(ctx) * sizeof(int32_t)));
  HIPCHK(hipMemsetAsync(ctx->d_win, 0, ntiles * nsens(ctx) * sizeof(int32_t), ctx->stream));
  return RGBDR_OK;
}

int rgbdr_set_inverse_calibration(rgbdr_ctx* ctx, int sensor, const rgbdr_lut* inv)
{
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (sensor < 0 || sensor >= nsens(ctx)) return ctx->fail(RGBDR_ERR_OUT_OF_RANGE, "sensor index out of range");
  if (!inv || !inv->data) return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, "null inverse calibration volume");
  for (int a = 0; a < 3; ++a)
    if (inv->res[a] < 1) return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, "empty inverse calibration volume");
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

static int read_lut_file(rgbdr_ctx* ctx, const char* path, size_t rec_bytes, rgbdr_lut* lut, std::vector<char>* buf)
{
  FILE* f = std::fopen(path, "rb");
  if (!f) return ctx->fail(RGBDR_ERR_IO, std::string("cannot open ") + path);
  bool ok = std::fread(lut->res, 4, 3, f) == 3 && std::fread(lut->depth_limits, 4, 2, f) == 2;
  size_t n = 0;
  if (ok) {
    n = (size_t)lut->res[0] * lut->res[1] * lut->res[2] * rec_bytes;
    buf->resize(n);
    ok = std::fread(buf->data(), 1, n, f) == n;
  }
  std::fclose(f);
  if (!ok) return ctx->fail(RGBDR_ERR_IO, std::string("short read from ") + path);
  lut->data = buf->data();
  return RGBDR_OK;
}

int rgbdr_load_calibration_files(rgbdr_ctx* ctx, int sensor, const char* pxyz, const char* puv, const char* pinv)
{
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

int rgbdr_synth_inverse_calibration(rgbdr_ctx* ctx, int sensor, const rgbdr_pinhole* cam)
{
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (sensor < 0 || sensor >= nsens(ctx)) return ctx->fail(RGBDR_ERR_OUT_OF_RANGE, "sensor index out of range");
  if (!cam) return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, "null pinhole");
  HIPCHK(hipSetDevice(ctx->device));
  const rgbdr_geometry& g = ctx->geo;
  const uint32_t r[3] = {(uint32_t)g.res_volume[0], (uint32_t)g.res_volume[1], (uint32_t)g.res_volume[2]};
  if (!lut_is_one_to_one(r, g.res_volume))
    return ctx->fail(RGBDR_ERR_STATE, "synthetic inverse LUT needs a grid whose voxel centres hit texel centres exactly");
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
  for (int a = 0; a < 3; ++a) ctx->inv_res[sensor][a] = r[a];
  (void)hipFree(ctx->d_lut_generic[sensor]);
  ctx->d_lut_generic[sensor] = nullptr;
  ctx->inv_tiled[sensor] = true;
  ctx->inv_set[sensor] = true;
  return RGBDR_OK;
}

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
  p->window = window < 1 ? 2 : window;
  p->sensor = sensor;
  p->N = ctx->cfg.num_sensors;
}

int rgbdr_compute_inverse_calibration(rgbdr_ctx* ctx, int sensor, int window)
{
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (sensor < 0 || sensor >= nsens(ctx)) return ctx->fail(RGBDR_ERR_OUT_OF_RANGE, "sensor index out of range");
  if (!ctx->have_calib[sensor]) return ctx->fail(RGBDR_ERR_STATE, "compute_inverse_calibration before set_calibration");
  if (window > 8) return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, "window radius must be <= 8");
  HIPCHK(hipSetDevice(ctx->device));
  for (int i = 0; i < nsens(ctx); ++i)
    if (i != sensor && ctx->inv_set[i] && !ctx->inv_tiled[i])
      return ctx->fail(RGBDR_ERR_STATE, "other sensors hold file-layout inverse LUTs (RGBDR_FLAG_NO_RESAMPLE)");
  int rc = ensure_tiled_lut(ctx);
  if (rc != RGBDR_OK) return rc;
  const rgbdr_geometry& g = ctx->geo;
  InvertParams p;
  fill_invert_params(ctx, sensor, g.res_volume, window, &p);
  const LutExtent ext = lut_extent(ctx);
  p.z0 = ext.vz0;
  p.nz = ext.vz1 - ext.vz0;
  p.out_tiled = ext.dst;
  launch_invert_lut(p, ctx->stream);
  LAUNCHCHK("invert_lut");
  launch_tile_windows(ctx->d_lut_tiled, ctx->cfg.depth_w, ctx->cfg.depth_h,
                      g.tiles[0] * g.tiles[1] * (g.slab_tile_z1 - g.slab_tile_z0), sensor, nsens(ctx), ctx->d_win,
                      ctx->stream);
  LAUNCHCHK("tile_windows");
  HIPCHK(hipStreamSynchronize(ctx->stream));
  (void)hipFree(ctx->d_lut_generic[sensor]);
  ctx->d_lut_generic[sensor] = nullptr;
  for (int a = 0; a < 3; ++a) ctx->inv_res[sensor][a] = (uint32_t)g.res_volume[a];
  ctx->inv_tiled[sensor] = true;
  ctx->inv_resampled[sensor] = false;
  ctx->inv_set[sensor] = true;
  return RGBDR_OK;
}

int rgbdr_generate_inverse_lut(rgbdr_ctx* ctx, int sensor, const uint32_t res[3], int window, float* dst)
{
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (sensor < 0 || sensor >= nsens(ctx)) return ctx->fail(RGBDR_ERR_OUT_OF_RANGE, "sensor index out of range");
  if (!res || !dst || res[0] < 1 || res[1] < 1 || res[2] < 1) return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, "bad resolution / destination");
  if (!ctx->have_calib[sensor]) return ctx->fail(RGBDR_ERR_STATE, "generate_inverse_lut before set_calibration");
  if (window > 8) return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, "window radius must be <= 8");
  HIPCHK(hipSetDevice(ctx->device));
  const int32_t vr[3] = {(int32_t)res[0], (int32_t)res[1], (int32_t)res[2]};
  InvertParams p;
  fill_invert_params(ctx, sensor, vr, window, &p);
  const size_t row = (size_t)res[0] * res[1];
  const int chunk = 64;
  DevScratch tmp;
  HIPCHK(hipMalloc(&tmp.p, row * chunk * sizeof(float4)));
  for (int z = 0; z < (int)res[2]; z += chunk) {
    p.z0 = z;
    p.nz = z + chunk <= (int)res[2] ? chunk : (int)res[2] - z;
    p.out_linear = tmp.as<float4>();
    launch_invert_lut(p, ctx->stream);
    LAUNCHCHK("invert_lut");
    HIPCHK(hipMemcpyAsync(dst + row * 4 * (size_t)z, tmp.p, row * (size_t)p.nz * sizeof(float4), hipMemcpyDeviceToHost,
                          ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
  }
  return RGBDR_OK;
}

// ---------------------------------------------------------------------------
static int upload_common(rgbdr_ctx* ctx, const void* depth, const void* color, hipMemcpyKind kind)
{
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (!depth || !color) return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, "null frame pointer");
  HIPCHK(hipSetDevice(ctx->device));
  const size_t n = npx(ctx);
  const size_t ncol = (size_t)nsens(ctx) * ctx->cfg.color_w * ctx->cfg.color_h * 3;
  hipStream_t ps = ctx->pstream();
  if (kind == hipMemcpyDeviceToDevice && !ctx->cfg.compress_depth && !ctx->cfg.compress_rgb &&
      launch_copy_frames(depth, ctx->d_depth_raw, n * 4, color, ctx->d_color, ncol, ps)) {
    LAUNCHCHK("copy_frames");
    ctx->frame_uploaded = true;
    return RGBDR_OK;
  }
  if (ctx->cfg.compress_depth) {
    HIPCHK(hipMemcpyAsync(ctx->d_depth_u8, depth, n, kind, ps));
    launch_u8_to_unit(ctx->d_depth_u8, ctx->d_depth_raw, n, ps);
    LAUNCHCHK("u8_to_unit");
  } else {
    HIPCHK(hipMemcpyAsync(ctx->d_depth_raw, depth, n * 4, kind, ps));
  }
  if (ctx->cfg.compress_rgb) {
    const size_t layer = color_frame_bytes(ctx->cfg);
    HIPCHK(hipMemcpyAsync(ctx->d_color_dxt, color, layer * nsens(ctx), kind, ps));
    launch_decode_dxt(ctx->d_color_dxt, ctx->cfg.color_w, ctx->cfg.color_h, ctx->cfg.compress_rgb, nsens(ctx), layer,
                      ctx->d_color, ps);
    LAUNCHCHK("decode_dxt");
  } else {
    HIPCHK(hipMemcpyAsync(ctx->d_color, color, ncol, kind, ps));
  }
  ctx->frame_uploaded = true;
  return RGBDR_OK;
}

int rgbdr_upload_frame(rgbdr_ctx* ctx, const void* depth, const void* color)
{
  return upload_common(ctx, depth, color, hipMemcpyHostToDevice);
}
int rgbdr_upload_frame_device(rgbdr_ctx* ctx, const void* depth, const void* color)
{
  return upload_common(ctx, depth, color, hipMemcpyDeviceToDevice);
}

static size_t depth_frame_bytes_all(const rgbdr_ctx* ctx) { return npx(ctx) * (ctx->cfg.compress_depth ? 1 : 4); }
static size_t color_frame_bytes_all(const rgbdr_ctx* ctx) { return color_frame_bytes(ctx->cfg) * (size_t)nsens(ctx); }

int rgbdr_map_frame_buffer(rgbdr_ctx* ctx, void** depth, void** color, size_t* depth_bytes, size_t* color_bytes)
{
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

int rgbdr_upload_mapped_frame(rgbdr_ctx* ctx)
{
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  const int b = ctx->mapped_back;
  if (!ctx->h_depth[b]) return ctx->fail(RGBDR_ERR_STATE, "upload_mapped_frame before map_frame_buffer");
  int rc = upload_common(ctx, ctx->h_depth[b], ctx->h_color[b], hipMemcpyHostToDevice);  // page-locked: true async DMA
  if (rc != RGBDR_OK) return rc;
  HIPCHK(hipEventRecord(ctx->ev_mapped[b], ctx->pstream()));
  ctx->ev_mapped_rec[b] = true;
  ctx->mapped_back = 1 - b;  // swapBuffers
  return RGBDR_OK;
}

int rgbdr_clear_occupied_bricks(rgbdr_ctx* ctx)
{
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  ctx->clear_pending = true;  // performed by the first kernel of process_textures, or by whoever reads the counters first
  return RGBDR_OK;
}

// clearOccupiedBricks is deferred; anything that reads the counters before process_textures ran flushes it
static int flush_clear(rgbdr_ctx* ctx)
{
  if (!ctx->clear_pending) return RGBDR_OK;
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipMemsetAsync(ctx->d_counters, 0, (size_t)ctx->geo.num_bricks * sizeof(uint32_t), ctx->pstream()));
  ctx->clear_pending = false;
  return RGBDR_OK;
}

int rgbdr_process_textures(rgbdr_ctx* ctx)
{
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (!ctx->frame_uploaded) return ctx->fail(RGBDR_ERR_STATE, "process_textures before any frame was uploaded");
  for (int i = 0; i < nsens(ctx); ++i)
    if (!ctx->have_calib[i]) return ctx->fail(RGBDR_ERR_STATE, "process_textures before set_calibration of every sensor");
  HIPCHK(hipSetDevice(ctx->device));
  PreParams p{};
  p.N = nsens(ctx);
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
  p.brick_counters = ctx->d_counters;
  p.color = ctx->d_color;
  p.depth_morph = ctx->d_depth_morph;
  p.depth_rg = ctx->d_depth_rg;
  p.lab = ctx->d_lab;
  p.depth_b_rg = ctx->d_depth_b;
  p.silhouette = ctx->d_sil;
  p.normal = ctx->d_normal;
  p.quality = ctx->d_quality;
  // m_use_processed_depth: the filter pass reads the morph output instead of the
  // raw depth (NetKinectArray.cpp:287-289)
  p.depth_in = (ctx->cfg.flags & RGBDR_FLAG_PROCESSED) ? ctx->d_depth_morph : ctx->d_depth_raw;

  hipStream_t ps = ctx->pstream();
  const int w = ctx->wbuf;
  p.frame = ctx->frame_buf(w);
  if (ctx->pipelined() && ctx->ev_int_rec[w]) HIPCHK(hipStreamWaitEvent(ps, ctx->ev_int[w], 0));  // last reader of buffer w
  tbegin(ctx, "1preprocess", ps);
  tbegin(ctx, "morph", ps);
  launch_morph(p, ctx->d_depth_raw, ctx->d_depth_morph, ctx->clear_pending ? ctx->d_counters : nullptr,
               (unsigned)ctx->geo.num_bricks, ps);
  ctx->clear_pending = false;
  tend(ctx, "morph", ps);
  tbegin(ctx, "bilateral", ps);
  launch_pre_depth(p, ps);
  tend(ctx, "bilateral", ps);
  tbegin(ctx, "boundary", ps);
  launch_boundary(p, ps);
  tend(ctx, "boundary", ps);
  tbegin(ctx, "normal", ps);
  launch_normal(p, ps);
  tend(ctx, "normal", ps);
  tbegin(ctx, "quality", ps);
  launch_quality(p, ps);
  tend(ctx, "quality", ps);
  tend(ctx, "1preprocess", ps);
  LAUNCHCHK("process_textures");
  ctx->rbuf = w;
  if (ctx->pipelined()) {
    HIPCHK(hipEventRecord(ctx->ev_pre[w], ps));
    ctx->ev_pre_rec[w] = true;
  }
  ctx->textures_processed = true;
  return RGBDR_OK;
}

int rgbdr_update_occupied_bricks(rgbdr_ctx* ctx)
{
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  HIPCHK(hipSetDevice(ctx->device));
  { int rc_ = flush_clear(ctx); if (rc_ != RGBDR_OK) return rc_; }
  hipStream_t ps = ctx->pstream();
  const int w = ctx->rbuf;  // belongs to the frame process_textures just wrote
  tbegin(ctx, "bricks", ps);
  launch_update_occupied(ctx->d_counters, (uint32_t)ctx->geo.num_bricks, ctx->cfg.min_voxels_per_brick,
                         ctx->mask_buf(w), ctx->count_buf(w), ps);
  tend(ctx, "bricks", ps);
  LAUNCHCHK("update_occupied");
  if (ctx->pipelined()) {
    HIPCHK(hipEventRecord(ctx->ev_pre[w], ps));
    ctx->ev_pre_rec[w] = true;
  }
  ctx->mask_valid = true;
  return RGBDR_OK;
}

int rgbdr_set_occupied_bricks(rgbdr_ctx* ctx, const uint32_t* ids, size_t count)
{
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
  ctx->mask_valid = true;
  return RGBDR_OK;
}

int rgbdr_integrate(rgbdr_ctx* ctx)
{
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (!ctx->textures_processed) return ctx->fail(RGBDR_ERR_STATE, "integrate before process_textures");
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
    return ctx->fail(RGBDR_ERR_STATE, "inverse LUTs at 1:1 and at other resolutions cannot be mixed in one context yet");
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
  p.vbx = ctx->d_brick_tab;
  p.vby = p.vbx + g.res_volume[0];
  p.vbz = p.vby + g.res_volume[1];
  p.tbx = p.vbz + g.res_volume[2];
  p.tby = p.tbx + g.tiles[0];
  p.tbz = p.tby + g.tiles[1];
  p.ovx = ctx->bt.overflow[0];
  p.ovy = ctx->bt.overflow[1];
  p.bx = g.res_bricks[0];
  p.by = g.res_bricks[1];
  p.bz = g.res_bricks[2];
  p.tsdf = ctx->d_tsdf_owned;
  p.tile_list = ctx->d_tile_list;
  p.tile_count = ctx->d_tile_list + (size_t)p.TX * p.TY * p.ntz + ctx->tile_count_parity;
  p.tile_count_next = ctx->d_tile_list + (size_t)p.TX * p.TY * p.ntz + (1 - ctx->tile_count_parity);
  if (bricks && all_tiled) ctx->tile_count_parity ^= 1;
  p.tile_state = ctx->d_tile_state;
  const bool elide = !bricks && all_tiled && (ctx->cfg.flags & RGBDR_FLAG_ELIDE_STORES) != 0;
  p.elide_stores = elide ? 1 : 0;
  if ((!bricks && !elide) || !all_tiled) {  // sweeps that overwrite tiles without keeping tile_state
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
  if (ctx->pipelined() && ctx->ev_pre_rec[ctx->rbuf]) HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_pre[ctx->rbuf], 0));
  tbegin(ctx, "2integrate", ctx->stream);
  launch_integrate(p, all_tiled, ctx->stream);
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
  if (ctx->pipelined()) {
    HIPCHK(hipEventRecord(ctx->ev_int[ctx->rbuf], ctx->stream));
    ctx->ev_int_rec[ctx->rbuf] = true;
    ctx->wbuf = ctx->rbuf ^ 1;  // the next frame's pre_* chain may run while this sweep reads rbuf
  }
  return RGBDR_OK;
}

int rgbdr_step(rgbdr_ctx* ctx, const void* depth, const void* color)
{
  int rc = rgbdr_upload_frame(ctx, depth, color);
  if (rc == RGBDR_OK) rc = rgbdr_clear_occupied_bricks(ctx);
  if (rc == RGBDR_OK) rc = rgbdr_process_textures(ctx);
  if (rc == RGBDR_OK) rc = rgbdr_update_occupied_bricks(ctx);
  if (rc == RGBDR_OK) rc = rgbdr_integrate(ctx);
  return rc;
}

int rgbdr_sync(rgbdr_ctx* ctx)
{
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  return sync_all(ctx);
}

// ---------------------------------------------------------------------------
int rgbdr_set_voxel_size(rgbdr_ctx* ctx, float size)
{
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

int rgbdr_set_brick_size(rgbdr_ctx* ctx, float size)
{
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

int rgbdr_set_tsdf_limit(rgbdr_ctx* ctx, float limit)
{
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (!(limit > 0.0f)) return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, "tsdf limit must be > 0");
  ctx->cfg.tsdf_limit = limit;
  return bump_clear_epoch(ctx);  // tiles cleared to the old -limit no longer count as cleared
}

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
{
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  int rc = sync_all(ctx);
  if (rc != RGBDR_OK) return rc;
  ctx->wbuf = ctx->rbuf;  // keep reading what was written last
  ctx->ev_pre_rec[0] = ctx->ev_pre_rec[1] = ctx->ev_int_rec[0] = ctx->ev_int_rec[1] = false;
  return set_flag(ctx, RGBDR_FLAG_PIPELINE, on);
}
int rgbdr_set_elide_stores(rgbdr_ctx* ctx, int on) { return set_flag(ctx, RGBDR_FLAG_ELIDE_STORES, on); }
int rgbdr_filter_textures(rgbdr_ctx* ctx, int on) { return set_flag(ctx, RGBDR_FLAG_FILTER, on); }
int rgbdr_use_processed_depths(rgbdr_ctx* ctx, int on) { return set_flag(ctx, RGBDR_FLAG_PROCESSED, on); }
int rgbdr_refine_boundary(rgbdr_ctx* ctx, int on) { return set_flag(ctx, RGBDR_FLAG_REFINE, on); }

int rgbdr_set_min_voxels_per_brick(rgbdr_ctx* ctx, uint32_t n)
{
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  ctx->cfg.min_voxels_per_brick = n;
  return RGBDR_OK;
}

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
{
  if (!ctx || !out) return RGBDR_ERR_INVALID_ARGUMENT;
  *out = ctx->geo;
  return RGBDR_OK;
}

int rgbdr_get_camera_position(const rgbdr_ctx* ctx, int sensor, float out[3])
{
  if (!ctx || !out) return RGBDR_ERR_INVALID_ARGUMENT;
  if (sensor < 0 || sensor >= ctx->cfg.num_sensors || !ctx->have_calib[sensor]) return RGBDR_ERR_OUT_OF_RANGE;
  std::memcpy(out, ctx->cam_pos[sensor], 12);
  return RGBDR_OK;
}

// ---------------------------------------------------------------------------
int rgbdr_readback_tsdf(rgbdr_ctx* ctx, float* dst)
{
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

int rgbdr_readback_image(rgbdr_ctx* ctx, int which, int sensor, float* dst)
{
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

int rgbdr_device_image(rgbdr_ctx* ctx, int which, int sensor, rgbdr_image_device_view* out)
{
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
      v.width = ctx->cfg.color_w;
      v.height = ctx->cfg.color_h;
      v.channels = 3;
      v.element_bytes = 1;
      v.ptr = ctx->d_color + (size_t)ctx->cfg.color_w * ctx->cfg.color_h * 3 * sensor;
      break;
    default: return ctx->fail(RGBDR_ERR_OUT_OF_RANGE, "unknown image id");
  }
  if (f) v.ptr = (void*)(f + px * v.channels * sensor);
  v.stream = ctx->pstream();
  *out = v;
  return RGBDR_OK;
}

int rgbdr_readback_inverse_calibration(rgbdr_ctx* ctx, int sensor, int z0, int z1, float* dst)
{
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

int rgbdr_readback_color(rgbdr_ctx* ctx, int sensor, uint8_t* dst)
{
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (!dst) return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, "null destination");
  if (sensor < 0 || sensor >= nsens(ctx)) return ctx->fail(RGBDR_ERR_OUT_OF_RANGE, "sensor index out of range");
  { int rc_ = sync_all(ctx); if (rc_ != RGBDR_OK) return rc_; }
  const size_t img = (size_t)ctx->cfg.color_w * ctx->cfg.color_h * 3;
  HIPCHK(hipMemcpyAsync(dst, ctx->d_color + img * sensor, img, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return RGBDR_OK;
}

int rgbdr_readback_brick_counters(rgbdr_ctx* ctx, uint32_t* dst)
{
  if (ctx) { int rc_ = flush_clear(ctx); if (rc_ != RGBDR_OK) return rc_; }
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (!dst) return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, "null destination");
  { int rc_ = sync_all(ctx); if (rc_ != RGBDR_OK) return rc_; }
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipMemcpyAsync(dst, ctx->d_counters, (size_t)ctx->geo.num_bricks * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return RGBDR_OK;
}

int rgbdr_get_occupied(rgbdr_ctx* ctx, uint32_t* ids, size_t capacity, size_t* count, float* ratio)
{
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (!count) return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, "null count");
  if (!ctx->mask_valid) return ctx->fail(RGBDR_ERR_STATE, "get_occupied before update_occupied_bricks");
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

int rgbdr_device_tsdf(rgbdr_ctx* ctx, rgbdr_tsdf_device_view* out)
{
  if (!ctx || !out) return RGBDR_ERR_INVALID_ARGUMENT;
  out->base = ctx->d_tsdf_base;
  out->owned = ctx->d_tsdf_owned;
  out->layer_bytes = ctx->layer_floats * sizeof(float);
  out->owned_layers = ctx->geo.slab_tile_z1 - ctx->geo.slab_tile_z0;
  out->halo_layers = ctx->halo;
  return RGBDR_OK;
}

int rgbdr_device_frame(rgbdr_ctx* ctx, int sensor, void** ptr)
{
  if (!ctx || !ptr) return RGBDR_ERR_INVALID_ARGUMENT;
  if (sensor < 0 || sensor >= nsens(ctx)) return ctx->fail(RGBDR_ERR_OUT_OF_RANGE, "sensor index out of range");
  *ptr = ctx->frame_buf(ctx->rbuf) + (size_t)ctx->cfg.depth_w * ctx->cfg.depth_h * sensor;
  return RGBDR_OK;
}

// d_view holds, per pixel: rgba (4), depth (1), samples (1), depth peels (4), first-hit index (1)
static int ensure_view_buffers(rgbdr_ctx* ctx, size_t npix)
{
  if (ctx->view_pixels >= npix) return RGBDR_OK;
  (void)hipFree(ctx->d_view);
  ctx->d_view = nullptr;
  ctx->view_pixels = 0;
  HIPCHK(hipMalloc((void**)&ctx->d_view, npix * 11 * sizeof(float)));
  ctx->view_pixels = npix;
  return RGBDR_OK;
}

static void mat4_product(const float* a, const float* b, float* o)  // glm association, column-major
{
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) o[4 * c + r] = a[r] * b[4 * c] + a[4 + r] * b[4 * c + 1] + a[8 + r] * b[4 * c + 2] + a[12 + r] * b[4 * c + 3];
}

// ReconIntegration::drawDepthLimits into the peel image of the view buffers
static int draw_depth_limits(rgbdr_ctx* ctx, const rgbdr_view* v, float4* out)
{
  { int rc_ = flush_clear(ctx); if (rc_ != RGBDR_OK) return rc_; }
  if (!ctx->mask_valid) return ctx->fail(RGBDR_ERR_STATE, "depth limits before update_occupied_bricks");
  PeelParams p{};
  mat4_product(v->projection, v->modelview, p.pmv);
  std::memcpy(p.modelview_inv, v->modelview_inv, 64);
  std::memcpy(p.img_to_eye, v->img_to_eye, 64);
  p.width = v->width;
  p.height = v->height;
  for (int a = 0; a < 3; ++a) {
    p.bbox_min[a] = ctx->cfg.bbox_min[a];
    p.res_bricks[a] = ctx->geo.res_bricks[a];
  }
  p.brick_size = ctx->geo.brick_size;
  p.counters = ctx->d_counters;
  p.mask = ctx->mask_buf(ctx->rbuf);
  p.out = out;
  tbegin(ctx, "brickdraw", ctx->stream);
  launch_depth_peels(p, ctx->stream);
  tend(ctx, "brickdraw", ctx->stream);
  LAUNCHCHK("depth_peels");
  return RGBDR_OK;
}

int rgbdr_draw_depth_limits(rgbdr_ctx* ctx, const rgbdr_view* v, float* peels)
{
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (!v || v->width < 1 || v->height < 1) return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, "bad view");
  { int rc_ = sync_all(ctx); if (rc_ != RGBDR_OK) return rc_; }
  const size_t npix = (size_t)v->width * v->height;
  int rc = ensure_view_buffers(ctx, npix);
  if (rc != RGBDR_OK) return rc;
  float4* out = (float4*)(ctx->d_view + npix * 6);
  rc = draw_depth_limits(ctx, v, out);
  if (rc != RGBDR_OK) return rc;
  if (peels) HIPCHK(hipMemcpyAsync(peels, out, npix * 16, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return RGBDR_OK;
}

// uniforms + resident data of the ray-marcher for `v`; runs the depth peels when asked
static int prepare_raymarch(rgbdr_ctx* ctx, const rgbdr_view* v, RaymarchParams* pp)
{
  if (!v || v->width < 1 || v->height < 1) return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, "bad view");
  if (v->shade_mode < 0 || v->shade_mode > 3) return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, "shade_mode must be 0..3");
  if (!ctx->integrated) return ctx->fail(RGBDR_ERR_STATE, "raymarch before integrate");
  const int N = nsens(ctx);
  bool tiled = true;
  for (int i = 0; i < N; ++i) tiled = tiled && ctx->inv_tiled[i];
  { int rc_ = sync_all(ctx); if (rc_ != RGBDR_OK) return rc_; }
  const size_t npix = (size_t)v->width * v->height;
  {
    int rc_ = ensure_view_buffers(ctx, npix);
    if (rc_ != RGBDR_OK) return rc_;
  }
  const rgbdr_geometry& g = ctx->geo;
  if (ctx->cfg.slab_count > 1) {
    const int rows = (int)std::ceil(ctx->cfg.tsdf_limit * (float)g.res_volume[2]) + 2;
    if (rows > ctx->halo * kTile) return ctx->fail(RGBDR_ERR_STATE, "tsdf_limit grew beyond what the slab halo covers; recreate the context");
  }
  RaymarchParams& p = *pp;
  p = RaymarchParams{};
  p.skip_space = v->skip_space ? 1 : 0;
  p.peels = (const float4*)(ctx->d_view + npix * 6);
  if (p.skip_space) {  // m_skip_space && m_use_bricks: drawDepthLimits first (recon_integration.cpp:153-156)
    int rc_ = draw_depth_limits(ctx, v, (float4*)(ctx->d_view + npix * 6));
    if (rc_ != RGBDR_OK) return rc_;
  }
  std::memcpy(p.projection, v->projection, 64);
  std::memcpy(p.normal_matrix, v->normal_matrix, 64);
  std::memcpy(p.gl_normal_matrix_inv, v->gl_normal_matrix_inv, 64);
  std::memcpy(p.vol_to_world_inv, v->vol_to_world_inv, 64);
  std::memcpy(p.modelview_inv, v->modelview_inv, 64);
  std::memcpy(p.img_to_eye, v->img_to_eye, 64);
  // gl_ModelViewMatrix * vol_to_world, evaluated once (the shader forms it per fragment, :123)
  mat4_product(v->modelview, v->vol_to_world, p.mv_vol_to_world);
  std::memcpy(p.camera_pos, v->camera_pos, 12);
  p.width = v->width;
  p.height = v->height;
  p.shade_mode = v->shade_mode;
  p.limit = ctx->cfg.tsdf_limit;
  p.N = N;
  p.W = ctx->cfg.depth_w;
  p.H = ctx->cfg.depth_h;
  p.Wc = ctx->cfg.color_w;
  p.Hc = ctx->cfg.color_h;
  p.X = g.res_volume[0];
  p.Y = g.res_volume[1];
  p.Z = g.res_volume[2];
  p.TX = g.tiles[0];
  p.TY = g.tiles[1];
  p.tz_alloc0 = g.slab_tile_z0 - ctx->halo;
  p.own_z0 = g.slab_voxel_z0;
  p.own_z1 = g.slab_voxel_z1;
  p.res_z0 = (g.slab_tile_z0 - ctx->halo) * kTile < 0 ? 0 : (g.slab_tile_z0 - ctx->halo) * kTile;
  p.res_z1 = (g.slab_tile_z1 + ctx->halo) * kTile > g.res_volume[2] ? g.res_volume[2] : (g.slab_tile_z1 + ctx->halo) * kTile;
  p.tsdf = ctx->d_tsdf_base;
  p.lut_tiled = tiled ? ctx->d_lut_tiled_base : nullptr;
  const size_t img = (size_t)p.W * p.H;
  for (int i = 0; i < N; ++i) {
    p.lut[i] = ctx->d_lut_generic[i];
    p.rx[i] = (int)ctx->inv_res[i][0];
    p.ry[i] = (int)ctx->inv_res[i][1];
    p.rz[i] = (int)ctx->inv_res[i][2];
    p.zoff[i] = ctx->zoff[i];
    p.cv_uv[i] = ctx->d_cv_uv[i];
    for (int a = 0; a < 3; ++a) p.uv_res[i][a] = (int)ctx->uv_res[i][a];
    p.frame[i] = ctx->frame_buf(ctx->rbuf) + img * i;
  }
  p.color = ctx->d_color;
  p.out_color = (float4*)ctx->d_view;
  p.out_depth = ctx->d_view + npix * 4;
  p.out_samples = ctx->d_view + npix * 5;
  p.khit = (int*)(ctx->d_view + npix * 10);
  ctx->view_w = v->width;
  ctx->view_h = v->height;
  return RGBDR_OK;
}

static int download_view(rgbdr_ctx* ctx, const RaymarchParams& p, float* color, float* depth, float* num_samples)
{
  const size_t npix = (size_t)p.width * p.height;
  if (color) HIPCHK(hipMemcpyAsync(color, p.out_color, npix * 16, hipMemcpyDeviceToHost, ctx->stream));
  if (depth) HIPCHK(hipMemcpyAsync(depth, p.out_depth, npix * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (num_samples) HIPCHK(hipMemcpyAsync(num_samples, p.out_samples, npix * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return RGBDR_OK;
}

int rgbdr_raymarch(rgbdr_ctx* ctx, const rgbdr_view* v, float* color, float* depth, float* num_samples)
{
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (ctx->cfg.slab_count > 1)
    return ctx->fail(RGBDR_ERR_STATE, "a Z slab cannot ray-march alone: use rgbdr_raymarch_find / _shade across the slabs");
  RaymarchParams p;
  int rc = prepare_raymarch(ctx, v, &p);
  if (rc != RGBDR_OK) return rc;
  tbegin(ctx, "draw", ctx->stream);
  launch_raymarch(p, 0, ctx->stream);
  tend(ctx, "draw", ctx->stream);
  LAUNCHCHK("raymarch");
  return download_view(ctx, p, color, depth, num_samples);
}

int rgbdr_raymarch_find(rgbdr_ctx* ctx, const rgbdr_view* v, void** first_hit_device)
{
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  RaymarchParams p;
  int rc = prepare_raymarch(ctx, v, &p);
  if (rc != RGBDR_OK) return rc;
  tbegin(ctx, "draw", ctx->stream);
  launch_raymarch(p, 1, ctx->stream);
  tend(ctx, "draw", ctx->stream);
  LAUNCHCHK("raymarch_find");
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (first_hit_device) *first_hit_device = p.khit;
  return RGBDR_OK;
}

int rgbdr_raymarch_shade(rgbdr_ctx* ctx, const rgbdr_view* v, float* color, float* depth, float* num_samples)
{
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (!v || ctx->view_w != v->width || ctx->view_h != v->height)
    return ctx->fail(RGBDR_ERR_STATE, "raymarch_shade needs rgbdr_raymarch_find of the same view first");
  const int skip = v->skip_space;
  rgbdr_view v2 = *v;
  v2.skip_space = 0;  // the peels of the find pass are still in the view buffers
  RaymarchParams p;
  int rc = prepare_raymarch(ctx, &v2, &p);
  if (rc != RGBDR_OK) return rc;
  p.skip_space = skip ? 1 : 0;
  launch_raymarch(p, 2, ctx->stream);
  LAUNCHCHK("raymarch_shade");
  return download_view(ctx, p, color, depth, num_samples);
}

int rgbdr_fill_colors(rgbdr_ctx* ctx, float* color, float* depth)
{
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (ctx->view_w < 1 || !ctx->d_view) return ctx->fail(RGBDR_ERR_STATE, "fill_colors before raymarch");
  HIPCHK(hipSetDevice(ctx->device));
  FillLayout L;
  make_fill_layout(ctx->view_w, ctx->view_h, &L);
  const size_t na = (size_t)L.FW * L.H, npix = (size_t)L.W * L.H;
  const size_t need = na * 10 + npix * 5;  // two atlases (rgba + depth) + the filled frame
  if (ctx->fill_floats < need) {
    (void)hipFree(ctx->d_fill);
    ctx->d_fill = nullptr;
    ctx->fill_floats = 0;
    HIPCHK(hipMalloc((void**)&ctx->d_fill, need * sizeof(float)));
    ctx->fill_floats = need;
  }
  float4* ncol = (float4*)ctx->d_fill;
  float4* scol = (float4*)(ctx->d_fill + na * 4);
  float4* ocol = (float4*)(ctx->d_fill + na * 8);
  float* ndep = ctx->d_fill + na * 8 + npix * 4;
  float* sdep = ndep + na;
  float* odep = sdep + na;
  tbegin(ctx, "holefill", ctx->stream);
  launch_fill_colors(L, (const float4*)ctx->d_view, ctx->d_view + npix * 4, ncol, ndep, scol, sdep, ocol, odep,
                     ctx->stream);
  tend(ctx, "holefill", ctx->stream);
  LAUNCHCHK("fill_colors");
  if (color) HIPCHK(hipMemcpyAsync(color, ocol, npix * 16, hipMemcpyDeviceToHost, ctx->stream));
  if (depth) HIPCHK(hipMemcpyAsync(depth, odep, npix * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return RGBDR_OK;
}

int rgbdr_upload_view_frame(rgbdr_ctx* ctx, int width, int height, const float* color, const float* depth)
{
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (width < 1 || height < 1 || !color || !depth) return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, "bad view frame");
  HIPCHK(hipSetDevice(ctx->device));
  { int rc_ = sync_all(ctx); if (rc_ != RGBDR_OK) return rc_; }
  const size_t npix = (size_t)width * height;
  int rc = ensure_view_buffers(ctx, npix);
  if (rc != RGBDR_OK) return rc;
  HIPCHK(hipMemcpyAsync(ctx->d_view, color, npix * 16, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(ctx->d_view + npix * 4, depth, npix * 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ctx->view_w = width;
  ctx->view_h = height;
  return RGBDR_OK;
}

int rgbdr_settle(rgbdr_ctx* ctx, float max_seconds, float* stream_ms)
{
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (!ctx->d_lut_tiled) return ctx->fail(RGBDR_ERR_STATE, "settle before the inverse LUTs were set");
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
    if ((double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec) > (double)max_seconds) break;
    struct timespec ts = {0, 50000000};
    nanosleep(&ts, nullptr);
  }
  if (stream_ms) *stream_ms = cur;
  return RGBDR_OK;
}

int rgbdr_get_arena_probe(const rgbdr_ctx* ctx, float ms[16], int* trials, int* chosen)
{
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (ms) std::memcpy(ms, ctx->arena_probe_ms, sizeof(ctx->arena_probe_ms));
  if (trials) *trials = ctx->arena_trials;
  if (chosen) *chosen = ctx->arena_chosen;
  return RGBDR_OK;
}

int rgbdr_halo_staging(rgbdr_ctx* ctx, int buffer, void** lo, void** hi, size_t* bytes)
{
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (buffer < 0 || buffer > 1) return ctx->fail(RGBDR_ERR_OUT_OF_RANGE, "halo staging buffer must be 0 or 1");
  if (ctx->halo <= 0) return ctx->fail(RGBDR_ERR_STATE, "halo staging needs a Z-slab context (slab_count > 1)");
  HIPCHK(hipSetDevice(ctx->device));
  const size_t face = ctx->layer_floats * (size_t)ctx->halo * sizeof(float);
  for (int f = 0; f < 2; ++f)
    if (!ctx->d_stage[buffer][f]) {
      HIPCHK(hipMalloc((void**)&ctx->d_stage[buffer][f], face));
      HIPCHK(hipMemsetAsync(ctx->d_stage[buffer][f], 0, face, ctx->stream));
    }
  if (lo) *lo = ctx->d_stage[buffer][0];
  if (hi) *hi = ctx->d_stage[buffer][1];
  if (bytes) *bytes = face;
  return RGBDR_OK;
}

int rgbdr_set_halo_staging(rgbdr_ctx* ctx, int buffer)
{
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (buffer > 1) return ctx->fail(RGBDR_ERR_OUT_OF_RANGE, "halo staging buffer must be 0, 1 or -1");
  if (buffer >= 0 && (!ctx->d_stage[buffer][0] || !ctx->d_stage[buffer][1]))
    return ctx->fail(RGBDR_ERR_STATE, "set_halo_staging before rgbdr_halo_staging of that buffer");
  ctx->stage_target = buffer < 0 ? -1 : buffer;
  return RGBDR_OK;
}

void* rgbdr_stream(rgbdr_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int rgbdr_set_stream(rgbdr_ctx* ctx, void* hip_stream)
{
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  { int rc_ = sync_all(ctx); if (rc_ != RGBDR_OK) return rc_; }
  ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
  return RGBDR_OK;
}

int rgbdr_set_timer_detail(rgbdr_ctx* ctx, int detail)
{
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  ctx->timer_detail = detail < 1 ? 0 : (detail < 2 ? 1 : 2);
  return RGBDR_OK;
}

int rgbdr_enable_timers(rgbdr_ctx* ctx, int on)
{
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  ctx->timers = on != 0;
  return RGBDR_OK;
}

int rgbdr_enable_timer_accumulation(rgbdr_ctx* ctx, int on)
{
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  ctx->accumulate = on != 0;
  if (on) ctx->timers = true;
  return RGBDR_OK;
}

int rgbdr_timer_stats(rgbdr_ctx* ctx, const char* name, uint64_t* total_ns, uint32_t* count)
{
  if (!ctx || !name || !total_ns || !count) return RGBDR_ERR_INVALID_ARGUMENT;
  *total_ns = 0;
  *count = 0;
  auto it = ctx->tm.find(name);
  if (it == ctx->tm.end()) return ctx->fail(RGBDR_ERR_OUT_OF_RANGE, std::string("no timer ") + name);
  // the intervals may have been recorded on either stream (pipelined mode: the pre_* timers
  // and "bricks" live on the second one)
  { int rc_ = sync_all(ctx); if (rc_ != RGBDR_OK) return rc_; }
  double total = 0.0;
  auto& pend = it->second.pending;
  while (!pend.empty()) {
    const auto ev = pend.back();
    float ms = 0.0f;
    // an interval whose end was never recorded (begin without end) is dropped, not reported
    if (hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess) {
      total += (double)ms * 1.0e6;
      ++*count;
    } else {
      (void)hipGetLastError();
    }
    it->second.pool.push_back(ev);
    pend.pop_back();
  }
  *total_ns = (uint64_t)total;
  return RGBDR_OK;
}

int rgbdr_timer_ns(rgbdr_ctx* ctx, const char* name, uint64_t* ns)
{
  if (!ctx || !name || !ns) return RGBDR_ERR_INVALID_ARGUMENT;
  auto it = ctx->tm.find(name);
  if (it == ctx->tm.end() || !it->second.recorded) return ctx->fail(RGBDR_ERR_OUT_OF_RANGE, std::string("no timer ") + name);
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipEventSynchronize(it->second.b));
  float ms = 0.0f;
  HIPCHK(hipEventElapsedTime(&ms, it->second.a, it->second.b));
  *ns = (uint64_t)((double)ms * 1.0e6);
  return RGBDR_OK;
}

}  // extern "C"
