// api_skip.cpp -- RGBDR_FLAG_SKIP_BACKGROUND: the per-frame tables behind the verdicts of the full sweep
// (kernels_skip.hip, integrate_fold.cuh: integrate_group) and their diagnostics.
#include <cstring>

#include "context.hpp"

using namespace rgbdr;

static void skip_params(rgbdr_ctx* ctx, IntegrateParams& p)
{
  const int N = nsens(ctx);
  const rgbdr_geometry& g = ctx->geo;
  const size_t ntiles = (size_t)g.tiles[0] * g.tiles[1] * (g.slab_tile_z1 - g.slab_tile_z0);
  p.N = N;
  p.W = ctx->cfg.depth_w;
  p.H = ctx->cfg.depth_h;
  p.limit = ctx->cfg.tsdf_limit;
  p.win = ctx->d_win;
  p.win_dmin = reinterpret_cast<const float*>(ctx->d_win + ntiles * N);
  p.win_dmax = reinterpret_cast<const float*>(ctx->d_win + 2 * ntiles * N);
  p.win_ext = ctx->d_win + 3 * ntiles * N;
  p.bgmax = ctx->d_bgmax;
}

// RGBDR_FLAG_SKIP_BACKGROUND: the per-window bounds of the frame process_textures wrote last (once per frame),
// and the buffers of the sweep: verdict bytes (diagnostics), the list of tiles with an undecided sensor, its two
// counters, and a page-locked word the list length is copied to after every sweep (the next sweep's grid size)
int rgbdr::ensure_window_background(rgbdr_ctx* ctx)
{
  const int N = nsens(ctx);
  const rgbdr_geometry& g = ctx->geo;
  const size_t n = (size_t)N * 9 * (ctx->cfg.depth_w + 1) * (ctx->cfg.depth_h + 1);
  const size_t ntiles = (size_t)g.tiles[0] * g.tiles[1] * (g.slab_tile_z1 - g.slab_tile_z0);
  const size_t mask_bytes = (ntiles * N + 3) & ~(size_t)3;
  if (!ctx->d_bgmax) HIPCHK(hipMalloc((void**)&ctx->d_bgmax, n * sizeof(float)));
  if (!ctx->h_skip_count) {
    HIPCHK(hipHostMalloc((void**)&ctx->h_skip_count, sizeof(unsigned), hipHostMallocMapped));
    HIPCHK(hipHostGetDevicePointer((void**)&ctx->d_skip_count_host, ctx->h_skip_count, 0));
    *ctx->h_skip_count = 0xffffffffu;  // no estimate yet
  }
  if (ctx->skip_mask_tiles != ntiles) {
    (void)hipFree(ctx->d_skip_mask);
    (void)hipFree(ctx->d_skip_list);
    ctx->d_skip_mask = nullptr;
    ctx->d_skip_list = nullptr;
    HIPCHK(hipMalloc((void**)&ctx->d_skip_mask, mask_bytes + sizeof(unsigned)));
    HIPCHK(hipMalloc((void**)&ctx->d_skip_list, (ntiles * (2 + N) + 2) * sizeof(unsigned)));  // behind the list: its two counters
    HIPCHK(hipMemsetAsync(ctx->d_skip_list + ntiles * (2 + N), 0, 2 * sizeof(unsigned), ctx->stream));
    ctx->skip_parity = 0;
    ctx->skip_mask_tiles = ntiles;
    ctx->bgmax_for = -1;
    *ctx->h_skip_count = 0xffffffffu;
  }
  if (ctx->bgmax_for == ctx->rbuf && ctx->skip_limit == ctx->cfg.tsdf_limit) return RGBDR_OK;
  launch_window_background(ctx->frame_buf(ctx->rbuf), ctx->cfg.depth_w, ctx->cfg.depth_h, N, ctx->d_bgmax, ctx->stream);
  LAUNCHCHK("window_background");
  ctx->bgmax_for = ctx->rbuf;
  ctx->skip_limit = ctx->cfg.tsdf_limit;
  ctx->skip_mask_valid = false;
  return RGBDR_OK;
}

// the verdict byte per pair (diagnostics; the sweep's classifier takes the verdicts itself)
static int ensure_skip_mask(rgbdr_ctx* ctx)
{
  { int rc_ = ensure_window_background(ctx); if (rc_ != RGBDR_OK) return rc_; }
  if (ctx->skip_mask_valid) return RGBDR_OK;
  IntegrateParams p{};
  skip_params(ctx, p);
  launch_skip_mask(p, (unsigned)(ctx->skip_mask_tiles * nsens(ctx)), ctx->d_skip_mask, ctx->stream);
  LAUNCHCHK("skip_mask");
  ctx->skip_mask_valid = true;
  return RGBDR_OK;
}

// the sweep itself: classifier (fills the tiles every sensor has a verdict for) + one block per listed tile
int rgbdr::skip_sweep(rgbdr_ctx* ctx, IntegrateParams& p)
{
  { int rc_ = ensure_window_background(ctx); if (rc_ != RGBDR_OK) return rc_; }
  const size_t ntiles = ctx->skip_mask_tiles;
  skip_params(ctx, p);
  unsigned* counts = ctx->d_skip_list + ntiles * (2 + (size_t)nsens(ctx));
  p.skip_list = ctx->d_skip_list;
  p.skip_count = counts + ctx->skip_parity;
  p.skip_count_next = counts + (1 - ctx->skip_parity);
  ctx->skip_parity ^= 1;
  // grid: the list length of the previous sweep (written to page-locked memory by that sweep; whatever it holds right
  // now is an estimate -- the blocks stride over the list) plus a margin, at least a machine-full of blocks
  const unsigned est = *(volatile unsigned*)ctx->h_skip_count;
  unsigned blocks = (unsigned)ntiles;
  if (est != 0xffffffffu) {
    const unsigned long long want = (unsigned long long)est + est / 8 + 256;
    blocks = (unsigned)(want < 2560 ? 2560 : want);
    if (blocks > ntiles) blocks = (unsigned)ntiles;
  }
  p.skip_count_host = ctx->d_skip_count_host;  // written by the sweep itself: no copy on the stream
  launch_skip_sweep(p, blocks, ctx->stream);
  LAUNCHCHK("skip_sweep");
  return RGBDR_OK;
}

extern "C" {

int rgbdr_readback_skip_tables(rgbdr_ctx* ctx, int which, void* dst, size_t bytes)
try {
  if (!ctx || !dst) return RGBDR_ERR_INVALID_ARGUMENT;
  uint64_t a = 0, b = 0;
  int rc = rgbdr_skipped_pairs(ctx, &a, &b);  // state checks + tables of the current frame (verdict bytes included)
  if (rc != RGBDR_OK) return rc;
  const size_t npairs = (size_t)b;
  const void* src = nullptr;
  size_t n = 0;
  if (which == 0) {
    src = ctx->d_skip_mask;
    n = npairs;
  } else if (which == 1) {
    src = ctx->d_win;
    n = 4 * npairs * sizeof(int32_t);
  } else if (which == 2) {
    src = ctx->d_bgmax;
    n = (size_t)nsens(ctx) * 9 * (ctx->cfg.depth_w + 1) * (ctx->cfg.depth_h + 1) * sizeof(float);
  } else {
    return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, "which must be 0, 1 or 2");
  }
  if (bytes != n) return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, "destination size does not match the table");
  HIPCHK(hipMemcpy(dst, src, n, hipMemcpyDeviceToHost));
  return RGBDR_OK;
}
RGBDR_CONTAIN(ctx)

int rgbdr_skipped_pairs(rgbdr_ctx* ctx, uint64_t* skipped, uint64_t* total)
try {
  if (!ctx || !skipped || !total) return RGBDR_ERR_INVALID_ARGUMENT;
  if (!ctx->textures_processed) return ctx->fail(RGBDR_ERR_STATE, "skipped_pairs before process_textures");
  const int N = nsens(ctx);
  for (int i = 0; i < N; ++i)
    if (!ctx->inv_set[i] || !ctx->inv_tiled[i])
      return ctx->fail(RGBDR_ERR_STATE, "skipped_pairs needs a 1:1 or resampled inverse LUT of every sensor");
  HIPCHK(hipSetDevice(ctx->device));
  { int rc_ = sync_all(ctx); if (rc_ != RGBDR_OK) return rc_; }
  { int rc_ = ensure_skip_mask(ctx); if (rc_ != RGBDR_OK) return rc_; }
  const rgbdr_geometry& g = ctx->geo;
  const size_t ntiles = (size_t)g.tiles[0] * g.tiles[1] * (g.slab_tile_z1 - g.slab_tile_z0), npairs = ntiles * N;
  unsigned h = 0;
  if (ctx->cfg.tsdf_limit > 0.0f) {
    unsigned* count = reinterpret_cast<unsigned*>(ctx->d_skip_mask + ((npairs + 3) & ~(size_t)3));
    HIPCHK(hipMemsetAsync(count, 0, sizeof(unsigned), ctx->stream));
    launch_count_bytes(ctx->d_skip_mask, (unsigned)npairs, count, ctx->stream);
    LAUNCHCHK("count_bytes");
    HIPCHK(hipMemcpyAsync(&h, count, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
  }
  *skipped = h;
  *total = npairs;
  return RGBDR_OK;
}
RGBDR_CONTAIN(ctx)

}  // extern "C"
