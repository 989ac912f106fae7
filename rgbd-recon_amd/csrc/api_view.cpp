// api_view.cpp -- consumers of the volume: ReconIntegration::drawDepthLimits,
// ::draw (ray-march, whole volume and Z slabs) and ::fillColors.
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>

#include "context.hpp"

using namespace rgbdr;

// a viewport like the reference's window (glViewport): bounded so that pixel counts times bytes per pixel cannot wrap
static bool view_size_ok(int w, int h) { return w >= 1 && h >= 1 && w <= 32768 && h <= 32768; }
// The march is bounded by the unit cube only for finite uniforms: tsdf_raymarch.fs:86 converts ceil(|t_far - t_near|) to a
// sample count, and a NaN / infinite camera makes that the largest unsigned -- four billion samples per ray, a kernel that
// does not come back.  8 matrices + the camera position = 131 floats at the head of rgbdr_view.
static bool view_is_finite(const rgbdr_view* v)
{
  const float* f = v->modelview;
  for (int i = 0; i < 8 * 16 + 3; ++i)
    if (!std::isfinite(f[i])) return false;
  return true;
}
static_assert(offsetof(rgbdr_view, camera_pos) == 8 * 16 * sizeof(float) && offsetof(rgbdr_view, width) == (8 * 16 + 3) * sizeof(float),
              "rgbdr_view starts with eight 4x4 matrices and the camera position");

extern "C" {
// d_view holds, per pixel: rgba (4), depth (1), samples (1), first-hit index (1).  Only calls that write a whole new frame
// (ray-march, rgbdr_upload_view_frame) size it: growing it drops the frame it held.
static int ensure_view_buffers(rgbdr_ctx* ctx, size_t npix)
{
  if (ctx->view_pixels >= npix) return RGBDR_OK;
  if (ctx->fill_stream) HIPCHK(hipStreamSynchronize(ctx->fill_stream));  // (a side-stream fill reads the old buffers)
  ctx->fill_side = false;
  ctx->ev_fill_rec[0] = ctx->ev_fill_rec[1] = false;
  (void)hipFree(ctx->d_view_base);
  ctx->d_view_base = ctx->d_view = nullptr;
  ctx->view_pixels = 0;
  ctx->vbuf = 0;
  ctx->view_w = ctx->view_h = 0;   // (the callers set the new frame's size once it is written)
  ctx->filled_w = ctx->filled_h = 0;
  HIPCHK(hipMalloc((void**)&ctx->d_view_base, npix * 7 * 2 * sizeof(float)));  // two halves (context.hpp)
  ctx->d_view = ctx->d_view_base;
  ctx->view_pixels = npix;
  return RGBDR_OK;
}
// a hole filling on the side stream (rgbdr_draw of a pipelined context) is in flight: what follows on the context's stream and
// touches the frame, the atlas or the filled image comes after it
static int join_side_fill(rgbdr_ctx* ctx)
{
  if (!ctx->fill_side) return RGBDR_OK;
  HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_fill[ctx->vbuf], 0));
  ctx->fill_side = false;
  return RGBDR_OK;
}
// the depth peels of a viewport: RGBA32F per pixel, their own allocation (context.hpp)
static int ensure_peel_buffer(rgbdr_ctx* ctx, size_t npix)
{
  if (ctx->peel_pixels >= npix) return RGBDR_OK;
  (void)hipFree(ctx->d_peels);
  ctx->d_peels = nullptr;
  ctx->peel_pixels = 0;
  HIPCHK(hipMalloc((void**)&ctx->d_peels, npix * 4 * sizeof(float)));
  ctx->peel_pixels = npix;
  return RGBDR_OK;
}

static void mat4_product(const float* a, const float* b, float* o)  // glm association, column-major
{
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) o[4 * c + r] = a[r] * b[4 * c] + a[4 + r] * b[4 * c + 1] + a[8 + r] * b[4 * c + 2] + a[12 + r] * b[4 * c + 3];
}

// ReconIntegration::drawDepthLimits into the peel image of the view buffers
static int draw_depth_limits(rgbdr_ctx* ctx, const rgbdr_view* v, float4* out, bool with_empty_tiles = false)
{
  { int rc_ = flush_clear(ctx); if (rc_ != RGBDR_OK) return rc_; }
  if (!ctx->mask_valid) return ctx->fail(RGBDR_ERR_STATE, "depth limits before update_occupied_bricks");
  { int rc_ = materialise_mask(ctx); if (rc_ != RGBDR_OK) return rc_; }
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
  p.counters = ctx->counters_cur();
  p.mask = ctx->mask_buf(ctx->rbuf);
  size_t supers = 1;
  for (int a = 0; a < 3; ++a) {
    p.res_super[a] = (p.res_bricks[a] + 3) / 4;
    supers *= (size_t)p.res_super[a];
  }
  (void)supers;
  const size_t bricks = (size_t)ctx->geo.num_bricks;
  if (ctx->peel_near_cap < bricks) {
    (void)hipFree(ctx->d_peel_near);
    ctx->d_peel_near = nullptr;
    ctx->peel_near_cap = 0;
    HIPCHK(hipMalloc((void**)&ctx->d_peel_near, bricks));
    ctx->peel_near_cap = bricks;
  }
  p.cells = ctx->d_peel_near;
  p.out = out;
  if (with_empty_tiles) p.empty_tiles = {ctx->d_tile_state, ctx->clear_epoch, ctx->geo.tiles[0], ctx->geo.tiles[1], ctx->geo.tiles[2], ctx->d_empty_tiles};
  tbegin(ctx, "brickdraw", ctx->stream);
  launch_depth_peels(p, ctx->stream);
  tend(ctx, "brickdraw", ctx->stream);
  LAUNCHCHK("depth_peels");
  return RGBDR_OK;
}

int rgbdr_draw_depth_limits(rgbdr_ctx* ctx, const rgbdr_view* v, float* peels)
try {
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (!v || !view_size_ok(v->width, v->height)) return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, "bad view (null, or not 1 ... 32768 pixels each way)");
  if (!view_is_finite(v)) return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, "bad view: a matrix or the camera position holds a NaN or an infinity");
  { int rc_ = sync_all(ctx); if (rc_ != RGBDR_OK) return rc_; }
  const size_t npix = (size_t)v->width * v->height;
  int rc = ensure_peel_buffer(ctx, npix);
  if (rc != RGBDR_OK) return rc;
  float4* out = (float4*)ctx->d_peels;
  rc = draw_depth_limits(ctx, v, out);
  if (rc != RGBDR_OK) return rc;
  if (peels) HIPCHK(hipMemcpyAsync(peels, out, npix * 16, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return RGBDR_OK;
}
RGBDR_CONTAIN(ctx)

// uniforms + resident data of the ray-marcher for `v`; runs the depth peels when asked
// `in_stream_order`: the caller enqueues behind the frame's passes on the context's stream and returns without waiting
// (rgbdr_draw); contexts with side streams (pipelined chain, halo / gather transfers) are drained first all the same
static int prepare_raymarch(rgbdr_ctx* ctx, const rgbdr_view* v, RaymarchParams* pp, bool in_stream_order = false)
{
  if (!v || !view_size_ok(v->width, v->height)) return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, "bad view (null, or not 1 ... 32768 pixels each way)");
  if (!view_is_finite(v)) return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, "bad view: a matrix or the camera position holds a NaN or an infinity");
  if (v->shade_mode < 0 || v->shade_mode > 3) return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, "shade_mode must be 0..3");
  if (!ctx->integrated) return ctx->fail(RGBDR_ERR_STATE, "raymarch before integrate");
  const int N = nsens(ctx);
  bool tiled = true;
  for (int i = 0; i < N; ++i) tiled = tiled && ctx->inv_tiled[i];
  if (!in_stream_order || ctx->gather_stream || ctx->halo_stream) {
    int rc_ = sync_all(ctx);
    if (rc_ != RGBDR_OK) return rc_;
  } else {
    HIPCHK(hipSetDevice(ctx->device));
    // a pipelined context: the frame's images, brick counters and mask come from the chain's stream (the sweep has waited
    // for them already if there was one); what the NEXT frame's chain overwrites while this pass runs is the other half of
    // every double buffer (ev_view_read orders the refills of this one, context.hpp)
    // (a wait or a record on a stream is a bubble of ~10 us between its kernels: none that is not needed)
    if (ctx->pipelined() && ctx->ev_pre_rec[ctx->rbuf] && ctx->pre_joined != ctx->pre_serial) {
      HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_pre[ctx->rbuf], 0));
      ctx->pre_joined = ctx->pre_serial;
    }
  }
  const size_t npix = (size_t)v->width * v->height;
  {
    int rc_ = ensure_view_buffers(ctx, npix);
    if (rc_ != RGBDR_OK) return rc_;
  }
  const rgbdr_geometry& g = ctx->geo;
  if (ctx->cfg.slab_count > 1) {
    const float frows = std::ceil(ctx->cfg.tsdf_limit * (float)g.res_volume[2]);   // (compared as a float: the limit may have grown without bound)
    const int rows = frows <= (float)kMaxRes ? (int)frows + 2 : kMaxRes + 2;
    if (rows > ctx->halo * kTile) return ctx->fail(RGBDR_ERR_STATE, "tsdf_limit grew beyond what the slab halo covers; recreate the context");
  }
  RaymarchParams& p = *pp;
  p = RaymarchParams{};
  p.skip_space = v->skip_space ? 1 : 0;
  // the sweep's record of the tiles that hold -limit throughout: samples there are not fetched (march_ahead)
  p.empty_bits = nullptr;
  p.empty_words = 0;
  bool want_empty_tiles = false;
  {
    const rgbdr_geometry& gg = ctx->geo;
    const size_t ntiles = (size_t)gg.tiles[0] * gg.tiles[1] * gg.tiles[2];
    const size_t words = (ntiles + 63) / 64 * 2;  // (a ballot of 64 tiles per store)
    if (ctx->cfg.slab_count == 1 && ctx->tile_states_kept && ctx->d_tile_state && words * 4 <= 48 * 1024 && !std::getenv("RGBDR_NO_EMPTY_TILES")) {
      if (ctx->empty_tiles_cap < words) {
        (void)hipFree(ctx->d_empty_tiles);
        ctx->d_empty_tiles = nullptr;
        ctx->empty_tiles_cap = 0;
        HIPCHK(hipMalloc((void**)&ctx->d_empty_tiles, words * 4));
        ctx->empty_tiles_cap = words;
      }
      want_empty_tiles = true;
      p.empty_bits = ctx->d_empty_tiles;
      p.empty_words = (int)words;
    }
  }
  if (p.skip_space) {  // m_skip_space && m_use_bricks: drawDepthLimits first (recon_integration.cpp:153-156)
    int rc_ = ensure_peel_buffer(ctx, npix);
    if (rc_ != RGBDR_OK) return rc_;
    rc_ = draw_depth_limits(ctx, v, (float4*)ctx->d_peels, want_empty_tiles);  // (the bitmap rides in the peels' first launch)
    if (rc_ != RGBDR_OK) return rc_;
  } else if (want_empty_tiles) {
    const rgbdr_geometry& gg = ctx->geo;
    launch_empty_tiles(ctx->d_tile_state, ctx->clear_epoch, gg.tiles[0], gg.tiles[1], gg.tiles[2], ctx->d_empty_tiles, ctx->stream);
    LAUNCHCHK("empty_tiles");
  }
  p.peels = (const float4*)ctx->d_peels;   // read with skip_space only
  // (developer knob, read per call: 1 = every ray still marching after a round goes to march_whole_wave, 2 = none does)
  const char* whole_wave = std::getenv("RGBDR_WHOLE_WAVE_MARCH");
  p.whole_wave = whole_wave ? std::atoi(whole_wave) : 0;
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
  // the shader samples the colour frames: RGB8, or -- a DXT upload nobody has asked the decoded frame of -- its blocks
  const int ch = ctx->color_of[ctx->rbuf];  // the colour half of the frame whose images the pass reads
  p.color = ctx->color_half(ch);
  p.color_dxt = nullptr;
  p.color_layer_bytes = 0;
  p.color_mode = ctx->cfg.compress_rgb;
  if (ctx->cfg.compress_rgb && !ctx->color_decoded[ch] && !std::getenv("RGBDR_DECODE_FOR_VIEW")) {
    const size_t blocks = (size_t)((ctx->cfg.color_w + 3) / 4) * ((ctx->cfg.color_h + 3) / 4);
    p.color_dxt = ctx->dxt_half(ch);
    p.color_layer_bytes = blocks * (ctx->cfg.compress_rgb == 1 ? 8 : 16);
  } else {
    int rc_ = ensure_color_decoded(ctx, ch);
    if (rc_ != RGBDR_OK) return rc_;
  }
  p.out_color = (float4*)ctx->d_view;
  p.out_depth = ctx->d_view + npix * 4;
  p.out_samples = ctx->d_view + npix * 5;
  p.khit = (int*)(ctx->d_view + npix * 6);
  ctx->view_w = v->width;
  ctx->view_h = v->height;
  ctx->filled_w = ctx->filled_h = 0;  // a new frame: the filled image no longer belongs to it
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
try {
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
RGBDR_CONTAIN(ctx)

int rgbdr_raymarch_find(rgbdr_ctx* ctx, const rgbdr_view* v, void** first_hit_device)
try {
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
RGBDR_CONTAIN(ctx)

int rgbdr_raymarch_shade(rgbdr_ctx* ctx, const rgbdr_view* v, float* color, float* depth, float* num_samples)
try {
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
RGBDR_CONTAIN(ctx)

// fillColors of the frame in the view buffers, enqueued on the context's stream; *ocol / *odep: the filled frame
static int fill_view_frame(rgbdr_ctx* ctx, float4** ocol_out, float** odep_out, hipStream_t s = nullptr)
{
  if (!s) {
    s = ctx->stream;
    int rc_ = join_side_fill(ctx);
    if (rc_ != RGBDR_OK) return rc_;
  }
  FillLayout L;
  make_fill_layout(ctx->view_w, ctx->view_h, &L);
  const size_t nb = fill_band_texels(L), npix = (size_t)L.W * L.H;
  const size_t need = (nb + npix) * 5 + 4;  // the atlas' LOD band (rgba + depth) + the filled frame
  if (ctx->fill_floats < need) {
    (void)hipFree(ctx->d_fill);
    ctx->d_fill = nullptr;
    ctx->fill_floats = 0;
    HIPCHK(hipMalloc((void**)&ctx->d_fill, need * 2 * sizeof(float)));  // two halves, like the view buffers (half vbuf: this frame's)
    ctx->fill_floats = need;
  }
  float* const fill = ctx->d_fill + (size_t)ctx->vbuf * ctx->fill_floats;
  float4* acol = (float4*)fill;
  float4* ocol = (float4*)(fill + nb * 4);
  float* adep = fill + (nb + npix) * 4;
  float* odep = adep + nb;
  if (ctx->fill_tab_w != L.W || ctx->fill_tab_h != L.H) {  // a new viewport size: its tap tables
    std::vector<int> xt, yt;
    FillTabs T;
    make_fill_tables(L, &xt, &yt, &T);
    HIPCHK(hipStreamSynchronize(ctx->stream));  // a fill in flight reads the old ones
    if (ctx->fill_stream) HIPCHK(hipStreamSynchronize(ctx->fill_stream));
    (void)hipFree(ctx->d_fill_tabs);
    ctx->d_fill_tabs = nullptr;
    ctx->fill_tab_w = ctx->fill_tab_h = 0;
    HIPCHK(hipMalloc((void**)&ctx->d_fill_tabs, (xt.size() + yt.size() + 4) * sizeof(int)));
    if (!xt.empty()) HIPCHK(hipMemcpy(ctx->d_fill_tabs, xt.data(), xt.size() * sizeof(int), hipMemcpyHostToDevice));
    if (!yt.empty()) HIPCHK(hipMemcpy(ctx->d_fill_tabs + xt.size(), yt.data(), yt.size() * sizeof(int), hipMemcpyHostToDevice));
    T.xt = (const int4*)ctx->d_fill_tabs;
    T.yt = (const int4*)(ctx->d_fill_tabs + xt.size());
    ctx->fill_tabs = T;
    ctx->fill_tab_w = L.W;
    ctx->fill_tab_h = L.H;
  }
  tbegin(ctx, "holefill", s);
  launch_fill_colors(L, ctx->fill_tabs, (const float4*)ctx->d_view, ctx->d_view + npix * 4, acol, adep, ocol, odep, s);
  tend(ctx, "holefill", s);
  LAUNCHCHK("fill_colors");
  ctx->filled_w = L.W;
  ctx->filled_h = L.H;
  *ocol_out = ocol;
  *odep_out = odep;
  return RGBDR_OK;
}

int rgbdr_fill_colors(rgbdr_ctx* ctx, float* color, float* depth)
try {
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (ctx->view_w < 1 || !ctx->d_view) return ctx->fail(RGBDR_ERR_STATE, "fill_colors before raymarch");
  HIPCHK(hipSetDevice(ctx->device));
  float4* ocol;
  float* odep;
  int rc = fill_view_frame(ctx, &ocol, &odep);
  if (rc != RGBDR_OK) return rc;
  const size_t npix = (size_t)ctx->view_w * ctx->view_h;
  if (color) HIPCHK(hipMemcpyAsync(color, ocol, npix * 16, hipMemcpyDeviceToHost, ctx->stream));
  if (depth) HIPCHK(hipMemcpyAsync(depth, odep, npix * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return RGBDR_OK;
}
RGBDR_CONTAIN(ctx)

int rgbdr_draw(rgbdr_ctx* ctx, const rgbdr_view* v, int fill_holes)
try {
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (ctx->cfg.slab_count > 1)
    return ctx->fail(RGBDR_ERR_STATE, "a Z slab cannot ray-march alone: use rgbdr_raymarch_find / _shade across the slabs");
  RaymarchParams p;
  // A pipelined context marches this frame into the other half of the view buffers and fills its holes on a stream of its
  // own: the sweep, the peels and the march of the NEXT frame run under that filling (seven small launches that wait on
  // each other).  The filled frame is then ordered on that stream: rgbdr_device_view_frame / rgbdr_readback_view_frame /
  // rgbdr_fill_colors make the context's stream wait for it.  (Not with the timers on: "3recon" brackets one stream.)
  const bool side_fill = ctx->pipelined() && fill_holes && !ctx->timers && ctx->d_view_base && v &&
                         (size_t)v->width * v->height <= ctx->view_pixels;
  if (side_fill) {
    if (!ctx->fill_stream) {
      HIPCHK(hipSetDevice(ctx->device));
      HIPCHK(hipStreamCreateWithFlags(&ctx->fill_stream, hipStreamNonBlocking));
      for (hipEvent_t& e : ctx->ev_fill) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    const int nb = ctx->vbuf ^ 1;
    if (ctx->ev_fill_rec[nb]) HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_fill[nb], 0));  // the fill that read that half: two frames ago
  } else {
    int rc_ = join_side_fill(ctx);
    if (rc_ != RGBDR_OK) return rc_;
  }
  const int vbuf_before = ctx->vbuf;
  const bool side_before = ctx->fill_side;
  if (side_fill) {
    ctx->vbuf ^= 1;
    ctx->d_view = ctx->d_view_base + (size_t)ctx->vbuf * ctx->view_pixels * 7;
    ctx->fill_side = false;  // (fills are in order on their stream: the one of the frame before needs no join)
  }
  tbegin(ctx, "3recon", ctx->stream);
  int rc = prepare_raymarch(ctx, v, &p, true);  // (drawDepthLimits inside, when the view asks for space skipping)
  if (rc != RGBDR_OK) {
    if (side_fill && ctx->d_view_base) {  // a view the library refuses leaves the last frame where it was
      ctx->vbuf = vbuf_before;
      ctx->d_view = ctx->d_view_base + (size_t)ctx->vbuf * ctx->view_pixels * 7;
      ctx->fill_side = side_before;
    }
    return rc;
  }
  tbegin(ctx, "draw", ctx->stream);
  launch_raymarch(p, 0, ctx->stream);
  tend(ctx, "draw", ctx->stream);
  LAUNCHCHK("raymarch");
  if (ctx->pipelined()) {
    // the pass has read the frame -- its images, counters, mask and colour: whoever refills those halves waits for this record
    // (context.hpp: the chain of the frame after the next, an upload into that colour half, the hole filling below)
    HIPCHK(hipEventRecord(ctx->ev_view_read[ctx->rbuf], ctx->stream));
    ctx->ev_view_rec[ctx->rbuf] = true;
    ctx->view_color[ctx->rbuf] = ctx->color_of[ctx->rbuf];
    ctx->int_unrecorded[ctx->rbuf] = false;  // ... and it covers the sweep of this frame, earlier on this stream
    ctx->draw_expected = true;
  }
  ctx->filled_w = ctx->filled_h = 0;
  if (fill_holes) {
    float4* ocol;
    float* odep;
    if (side_fill) {
      HIPCHK(hipStreamWaitEvent(ctx->fill_stream, ctx->ev_view_read[ctx->rbuf], 0));  // (recorded behind the march just above)
      rc = fill_view_frame(ctx, &ocol, &odep, ctx->fill_stream);
      if (rc != RGBDR_OK) return rc;
      HIPCHK(hipEventRecord(ctx->ev_fill[ctx->vbuf], ctx->fill_stream));
      ctx->ev_fill_rec[ctx->vbuf] = true;
      ctx->fill_side = true;
    } else {
      rc = fill_view_frame(ctx, &ocol, &odep);
      if (rc != RGBDR_OK) return rc;
    }
  }
  tend(ctx, "3recon", ctx->stream);
  return RGBDR_OK;
}
RGBDR_CONTAIN(ctx)

// where the displayed frame lives: the ray-marched one in the view buffers, the filled one behind the atlas band
// (`join`: what follows on the context's stream comes after a hole filling still in flight on its own stream)
static int view_frame_pointers(rgbdr_ctx* ctx, int filled, float** color, float** depth, bool join = true)
{
  if (ctx->view_w < 1 || !ctx->d_view) return ctx->fail(RGBDR_ERR_STATE, "no frame: nothing was ray-marched or uploaded");
  if (join) { int rc_ = join_side_fill(ctx); if (rc_ != RGBDR_OK) return rc_; }
  const size_t npix = (size_t)ctx->view_w * ctx->view_h;
  if (!filled) {
    *color = ctx->d_view;
    *depth = ctx->d_view + npix * 4;
    return RGBDR_OK;
  }
  if (ctx->filled_w != ctx->view_w || ctx->filled_h != ctx->view_h || !ctx->d_fill)
    return ctx->fail(RGBDR_ERR_STATE, "the frame in the view buffers has not been filled (rgbdr_fill_colors / rgbdr_draw with fill_holes)");
  FillLayout L;
  make_fill_layout(ctx->view_w, ctx->view_h, &L);
  const size_t nb = fill_band_texels(L);
  float* const fill = ctx->d_fill + (size_t)ctx->vbuf * ctx->fill_floats;
  *color = fill + nb * 4;
  *depth = fill + (nb + npix) * 4 + nb;
  return RGBDR_OK;
}

int rgbdr_device_view_frame(rgbdr_ctx* ctx, int filled, void** color, void** depth, int* width, int* height)
try {
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  float *c, *d;
  int rc = view_frame_pointers(ctx, filled, &c, &d);
  if (rc != RGBDR_OK) return rc;
  if (color) *color = c;
  if (depth) *depth = d;
  if (width) *width = ctx->view_w;
  if (height) *height = ctx->view_h;
  return RGBDR_OK;
}
RGBDR_CONTAIN(ctx)

int rgbdr_device_view_frame_async(rgbdr_ctx* ctx, int filled, void** color, void** depth, int* width, int* height, void** ready_event)
try {
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (!ready_event) return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, "null event pointer");
  float *c, *d;
  int rc = view_frame_pointers(ctx, filled, &c, &d, false);
  if (rc != RGBDR_OK) return rc;
  if (filled && ctx->fill_side) {
    *ready_event = (void*)ctx->ev_fill[ctx->vbuf];  // behind the filling on its own stream
  } else {  // the frame is ordered on the context's stream: an event behind what is enqueued there
    if (!ctx->ev_view_ready) HIPCHK(hipEventCreateWithFlags(&ctx->ev_view_ready, hipEventDisableTiming));
    HIPCHK(hipEventRecord(ctx->ev_view_ready, ctx->stream));
    *ready_event = (void*)ctx->ev_view_ready;
  }
  if (color) *color = c;
  if (depth) *depth = d;
  if (width) *width = ctx->view_w;
  if (height) *height = ctx->view_h;
  return RGBDR_OK;
}
RGBDR_CONTAIN(ctx)

int rgbdr_readback_view_frame(rgbdr_ctx* ctx, int filled, float* color, float* depth)
try {
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  float *c, *d;
  int rc = view_frame_pointers(ctx, filled, &c, &d);
  if (rc != RGBDR_OK) return rc;
  HIPCHK(hipSetDevice(ctx->device));
  const size_t npix = (size_t)ctx->view_w * ctx->view_h;
  if (color) HIPCHK(hipMemcpyAsync(color, c, npix * 16, hipMemcpyDeviceToHost, ctx->stream));
  if (depth) HIPCHK(hipMemcpyAsync(depth, d, npix * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return RGBDR_OK;
}
RGBDR_CONTAIN(ctx)

int rgbdr_upload_view_frame(rgbdr_ctx* ctx, int width, int height, const float* color, const float* depth)
try {
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (!view_size_ok(width, height) || !color || !depth) return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, "bad view frame (null, or not 1 ... 32768 pixels each way)");
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
  ctx->filled_w = ctx->filled_h = 0;
  return RGBDR_OK;
}
RGBDR_CONTAIN(ctx)

}  // extern "C"
