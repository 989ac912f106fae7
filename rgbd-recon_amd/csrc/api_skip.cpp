// api_skip.cpp -- RGBDR_FLAG_SKIP_BACKGROUND: the per-frame tables behind the verdicts of the full sweep
// (kernels_integrate.hip: integrate_group, k_window_background, k_skip_mask) and their diagnostics.
#include <cstring>

#include "context.hpp"

using namespace rgbdr;

// RGBDR_FLAG_SKIP_BACKGROUND: the per-window bounds of the frame process_textures wrote last (once per frame)
int rgbdr::ensure_window_background(rgbdr_ctx* ctx)
{
  const int N = nsens(ctx);
  const rgbdr_geometry& g = ctx->geo;
  const size_t n = (size_t)N * 9 * (ctx->cfg.depth_w + 1) * (ctx->cfg.depth_h + 1);
  const size_t ntiles = (size_t)g.tiles[0] * g.tiles[1] * (g.slab_tile_z1 - g.slab_tile_z0);
  const size_t mask_bytes = (ntiles * N + 3) & ~(size_t)3;
  if (!ctx->d_bgmax) HIPCHK(hipMalloc((void**)&ctx->d_bgmax, n * sizeof(float)));
  if (ctx->skip_mask_tiles != ntiles) {
    (void)hipFree(ctx->d_skip_mask);
    ctx->d_skip_mask = nullptr;
    HIPCHK(hipMalloc((void**)&ctx->d_skip_mask, mask_bytes + sizeof(unsigned)));
    ctx->skip_mask_tiles = ntiles;
    ctx->bgmax_for = -1;
  }
  if (ctx->bgmax_for == ctx->rbuf && ctx->skip_limit == ctx->cfg.tsdf_limit) return RGBDR_OK;
  launch_window_background(ctx->frame_buf(ctx->rbuf), ctx->cfg.depth_w, ctx->cfg.depth_h, N, ctx->d_bgmax, ctx->stream);
  IntegrateParams p{};
  p.N = N;
  p.W = ctx->cfg.depth_w;
  p.H = ctx->cfg.depth_h;
  p.limit = ctx->cfg.tsdf_limit;
  p.win = ctx->d_win;
  p.win_dmin = reinterpret_cast<const float*>(ctx->d_win + ntiles * N);
  p.win_dmax = reinterpret_cast<const float*>(ctx->d_win + 2 * ntiles * N);
  p.win_ext = ctx->d_win + 3 * ntiles * N;
  p.bgmax = ctx->d_bgmax;
  launch_skip_mask(p, (unsigned)(ntiles * N), ctx->d_skip_mask, ctx->stream);
  LAUNCHCHK("window_background");
  ctx->bgmax_for = ctx->rbuf;
  ctx->skip_limit = ctx->cfg.tsdf_limit;
  return RGBDR_OK;
}

extern "C" {

int rgbdr_readback_skip_tables(rgbdr_ctx* ctx, int which, void* dst, size_t bytes)
{
  if (!ctx || !dst) return RGBDR_ERR_INVALID_ARGUMENT;
  uint64_t a = 0, b = 0;
  int rc = rgbdr_skipped_pairs(ctx, &a, &b);  // state checks + tables of the current frame
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

int rgbdr_skipped_pairs(rgbdr_ctx* ctx, uint64_t* skipped, uint64_t* total)
{
  if (!ctx || !skipped || !total) return RGBDR_ERR_INVALID_ARGUMENT;
  if (!ctx->textures_processed) return ctx->fail(RGBDR_ERR_STATE, "skipped_pairs before process_textures");
  const int N = nsens(ctx);
  for (int i = 0; i < N; ++i)
    if (!ctx->inv_set[i] || !ctx->inv_tiled[i])
      return ctx->fail(RGBDR_ERR_STATE, "skipped_pairs needs a 1:1 or resampled inverse LUT of every sensor");
  HIPCHK(hipSetDevice(ctx->device));
  { int rc_ = sync_all(ctx); if (rc_ != RGBDR_OK) return rc_; }
  { int rc_ = ensure_window_background(ctx); if (rc_ != RGBDR_OK) return rc_; }
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

}  // extern "C"
