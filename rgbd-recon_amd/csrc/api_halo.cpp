// api_halo.cpp -- Z-slab halo: staging sets and the RCCL neighbour exchange.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>

#include <dlfcn.h>

#include "context.hpp"

using namespace rgbdr;

namespace {
// RCCL is bound at run time, from the copy that is already in the process (the host created the
// communicator with it) or else the system's: the library has no link-time dependency on RCCL, so
// single-GPU hosts need none.  Signatures: /opt/rocm/include/rccl/rccl.h (ncclFloat = 7).
struct Rccl {
  bool tried = false;
  void* lib = nullptr;
  int (*group_start)() = nullptr;
  int (*group_end)() = nullptr;
  int (*send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*all_gather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  int (*all_reduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*comm_count)(void*, int*) = nullptr;
  int (*comm_rank)(void*, int*) = nullptr;
  const char* (*error_string)(int) = nullptr;
  std::string why;
};

Rccl load_rccl()
{
  Rccl r;
  r.tried = true;
  const char* env = std::getenv("RGBDR_RCCL_LIB");
  const char* names[] = {env, "librccl.so.1", "librccl.so"};
  for (int pass = 0; pass < 2 && !r.lib; ++pass)  // pass 0: only a copy that is loaded already
    for (const char* n : names)
      if (n && !r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
  if (!r.lib) {
    r.why = "librccl.so.1 could not be loaded (set RGBDR_RCCL_LIB)";
    return r;
  }
  r.group_start = (int (*)())dlsym(r.lib, "ncclGroupStart");
  r.group_end = (int (*)())dlsym(r.lib, "ncclGroupEnd");
  r.send = (int (*)(const void*, size_t, int, int, void*, hipStream_t))dlsym(r.lib, "ncclSend");
  r.recv = (int (*)(void*, size_t, int, int, void*, hipStream_t))dlsym(r.lib, "ncclRecv");
  r.all_gather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(r.lib, "ncclAllGather");
  r.all_reduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(r.lib, "ncclAllReduce");
  r.comm_count = (int (*)(void*, int*))dlsym(r.lib, "ncclCommCount");
  r.comm_rank = (int (*)(void*, int*))dlsym(r.lib, "ncclCommUserRank");
  r.error_string = (const char* (*)(int))dlsym(r.lib, "ncclGetErrorString");
  if (!r.group_start || !r.group_end || !r.send || !r.recv) {
    r.why = "the RCCL library lacks ncclSend / ncclRecv / ncclGroupStart / ncclGroupEnd";
    r.lib = nullptr;
  }
  return r;
}

// bound once per process; a function-local static, so two contexts driven by two threads may ask at the same time
Rccl& rccl()
{
  static Rccl r = load_rccl();
  return r;
}

// One exchange on `st`: the h lowest owned tile layers (or staging buffer `lo`) go to peer_lo, the h
// highest (or `hi`) to peer_hi; the neighbours' faces arrive in the halo layers below / above the owned
// ones.  A tile layer range is contiguous in the tile-linear layout: one message per face, no packing.
int exchange_on(rgbdr_ctx* ctx, void* comm, int peer_lo, int peer_hi, int buffer, hipStream_t st)
{
  Rccl& r = rccl();
  if (!r.lib) return ctx->fail(RGBDR_ERR_STATE, r.why);
  const size_t face = ctx->layer_floats * (size_t)ctx->halo;
  const int owned = ctx->geo.slab_tile_z1 - ctx->geo.slab_tile_z0;
  const float* send_lo = buffer >= 0 ? ctx->d_stage[buffer][0] : ctx->d_tsdf_owned;
  const float* send_hi = buffer >= 0 ? ctx->d_stage[buffer][1] : ctx->d_tsdf_owned + ctx->layer_floats * (size_t)(owned - ctx->halo);
  float* recv_lo = ctx->d_tsdf_base;
  float* recv_hi = ctx->d_tsdf_owned + ctx->layer_floats * (size_t)owned;
  auto chk = [&](int rc, const char* what) {
    if (rc == 0) return 0;
    return ctx->fail(RGBDR_ERR_HIP, std::string(what) + ": " + (r.error_string ? r.error_string(rc) : "RCCL error"));
  };
  tbegin(ctx, "halo", st);
  int rc = chk(r.group_start(), "ncclGroupStart");
  if (rc == 0 && peer_lo >= 0) {
    rc = chk(r.send(send_lo, face, 7, peer_lo, comm, st), "ncclSend");
    if (rc == 0) rc = chk(r.recv(recv_lo, face, 7, peer_lo, comm, st), "ncclRecv");
  }
  if (rc == 0 && peer_hi >= 0) {
    rc = chk(r.send(send_hi, face, 7, peer_hi, comm, st), "ncclSend");
    if (rc == 0) rc = chk(r.recv(recv_hi, face, 7, peer_hi, comm, st), "ncclRecv");
  }
  const int rc_end = chk(r.group_end(), "ncclGroupEnd");
  tend(ctx, "halo", st);
  return rc != 0 ? rc : rc_end;
}

int check_slab(rgbdr_ctx* ctx, int buffer)
{
  if (ctx->halo <= 0) return ctx->fail(RGBDR_ERR_STATE, "the halo exchange needs a Z-slab context (slab_count > 1)");
  if (buffer > 1) return ctx->fail(RGBDR_ERR_OUT_OF_RANGE, "halo staging buffer must be 0, 1 or -1");
  if (buffer >= 0 && (!ctx->d_stage[buffer][0] || !ctx->d_stage[buffer][1]))
    return ctx->fail(RGBDR_ERR_STATE, "halo exchange from a staging set before rgbdr_halo_staging of that set");
  return RGBDR_OK;
}
}  // namespace

extern "C" {
int rgbdr_halo_staging(rgbdr_ctx* ctx, int buffer, void** lo, void** hi, size_t* bytes)
try {
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
RGBDR_CONTAIN(ctx)

int rgbdr_set_halo_staging(rgbdr_ctx* ctx, int buffer)
try {
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (buffer > 1) return ctx->fail(RGBDR_ERR_OUT_OF_RANGE, "halo staging buffer must be 0, 1 or -1");
  if (buffer >= 0 && (!ctx->d_stage[buffer][0] || !ctx->d_stage[buffer][1]))
    return ctx->fail(RGBDR_ERR_STATE, "set_halo_staging before rgbdr_halo_staging of that buffer");
  ctx->stage_target = buffer < 0 ? -1 : buffer;
  return RGBDR_OK;
}
RGBDR_CONTAIN(ctx)

int rgbdr_readback_tile_layers(rgbdr_ctx* ctx, int first, int count, float* dst)
try {
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (!dst) return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, "null destination");
  const int resident = ctx->geo.slab_tile_z1 - ctx->geo.slab_tile_z0 + 2 * ctx->halo;
  if (first < 0 || count < 1 || first + count > resident) return ctx->fail(RGBDR_ERR_OUT_OF_RANGE, "tile layers not resident");
  { int rc_ = sync_all(ctx); if (rc_ != RGBDR_OK) return rc_; }
  HIPCHK(hipMemcpy(dst, ctx->d_tsdf_base + ctx->layer_floats * (size_t)first, ctx->layer_floats * (size_t)count * sizeof(float),
                   hipMemcpyDeviceToHost));
  return RGBDR_OK;
}
RGBDR_CONTAIN(ctx)

int rgbdr_halo_exchange(rgbdr_ctx* ctx, void* nccl_comm, int peer_lo, int peer_hi, int buffer, void* hip_stream)
try {
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (!nccl_comm) return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, "null communicator");
  { int rc_ = check_slab(ctx, buffer); if (rc_ != RGBDR_OK) return rc_; }
  HIPCHK(hipSetDevice(ctx->device));
  return exchange_on(ctx, nccl_comm, peer_lo, peer_hi, buffer, hip_stream ? (hipStream_t)hip_stream : ctx->stream);
}
RGBDR_CONTAIN(ctx)

// ---- the managed form: staging sets, side stream and events owned by the context ---------------------
// Per step k (b = k mod 2), like rgbd-recon_amd/dist.py:HaloExchanger:
//   begin_step      context stream: [wait: transfer k-2 done, it read set b]; the sweep will fill set b
//   ... rgbdr_integrate ...
//   exchange_async  context stream: record staged_k;  side stream: wait staged_k, send set b / receive
//                   into the halo layers, record done_k
//   wait            context stream waits for the newest done event (consumers that sample across faces)
// The halo layers are only ever written by the side stream and only read after wait(), so a transfer
// overlaps the whole next frame.
int rgbdr_halo_begin_step(rgbdr_ctx* ctx)
try {
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  { int rc_ = check_slab(ctx, -1); if (rc_ != RGBDR_OK) return rc_; }
  HIPCHK(hipSetDevice(ctx->device));
  if (!ctx->halo_stream) {
    if (!ctx->side_cu_mask.empty())  // RGBDR_CU_SPLIT: RCCL's kernels run on the CUs set aside for the second stream
      HIPCHK(hipExtStreamCreateWithCUMask(&ctx->halo_stream, (uint32_t)ctx->side_cu_mask.size(), ctx->side_cu_mask.data()));
    else
      HIPCHK(hipStreamCreateWithFlags(&ctx->halo_stream, hipStreamNonBlocking));
    for (int b = 0; b < 2; ++b) {
      HIPCHK(hipEventCreateWithFlags(&ctx->ev_halo_done[b], hipEventDisableTiming));
      HIPCHK(hipEventCreateWithFlags(&ctx->ev_halo_staged[b], hipEventDisableTiming));
    }
  }
  const int b = ctx->halo_step & 1;
  { int rc_ = rgbdr_halo_staging(ctx, b, nullptr, nullptr, nullptr); if (rc_ != RGBDR_OK) return rc_; }
  if (ctx->halo_done_rec[b]) HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_halo_done[b], 0));
  if (ctx->peer) {  // copy-engine transport: the neighbours read staging set b themselves (api_peer.cpp)
    int rc_ = rgbdr::peer_begin_step(ctx, b, ctx->halo_step + 1);
    if (rc_ != RGBDR_OK) return rc_;
  }
  ctx->stage_target = b;
  ctx->halo_begun = true;
  ctx->halo_staged = false;  // set by the rgbdr_integrate that fills set b
  return RGBDR_OK;
}
RGBDR_CONTAIN(ctx)

int rgbdr_halo_exchange_async(rgbdr_ctx* ctx, void* nccl_comm, int peer_lo, int peer_hi)
try {
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (!nccl_comm) return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, "null communicator");
  if (!ctx->halo_begun) return ctx->fail(RGBDR_ERR_STATE, "rgbdr_halo_exchange_async before rgbdr_halo_begin_step + rgbdr_integrate");
  HIPCHK(hipSetDevice(ctx->device));
  const int b = ctx->halo_step & 1;
  { int rc_ = check_slab(ctx, b); if (rc_ != RGBDR_OK) return rc_; }
  if (ctx->stage_target != b || !ctx->halo_staged)
    return ctx->fail(RGBDR_ERR_STATE, "rgbdr_halo_exchange_async: no rgbdr_integrate has filled the staging set since rgbdr_halo_begin_step");
  ctx->halo_begun = false;
  ++ctx->halo_step;
  HIPCHK(hipEventRecord(ctx->ev_halo_staged[b], ctx->stream));
  HIPCHK(hipStreamWaitEvent(ctx->halo_stream, ctx->ev_halo_staged[b], 0));
  int rc = exchange_on(ctx, nccl_comm, peer_lo, peer_hi, b, ctx->halo_stream);
  if (rc != RGBDR_OK) return rc;
  HIPCHK(hipEventRecord(ctx->ev_halo_done[b], ctx->halo_stream));
  ctx->halo_done_rec[b] = true;
  ctx->halo_last = b;
  return RGBDR_OK;
}
RGBDR_CONTAIN(ctx)

// ---- the pre_* chain sharded by sensor over the ranks of a slab job ---------------------------------------
// Every rank of a Z-slab job needs the frame images of ALL sensors (a voxel may project anywhere), and with the chain
// run redundantly its time does not shrink with the number of GPUs while the sweep's does: at BASELINE configs[3]
// (8 sensors, 512^3 / 4) it is 0.14 of a rank's 0.70 ms.  With a shard, rank r runs the chain for sensors
// [r * n / k, (r + 1) * n / k) only; what the sweep (and the slab ray-march) read of a sensor is its packed 8-byte frame
// texel, so one all-gather of those (1.7 MB per 512 x 424 sensor) plus an all-reduce(sum) of the u32 brick counters (each
// rank counted its own sensors' pixels) completes the frame on every rank -- the alternative SURVEY.md 8(e) names.  Both
// run on the stream the chain ran on, so under RGBDR_FLAG_PIPELINE they overlap the sweep of the frame before.
int rgbdr_set_sensor_shard(rgbdr_ctx* ctx, int first, int count)
try {
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  const int N = nsens(ctx);
  const int old_first = ctx->shard_first, old_count = ctx->shard_count;
  if (count <= 0 || (first == 0 && count == N)) {  // back to every sensor
    ctx->shard_first = ctx->shard_count = 0;
  } else {
    if (first < 0 || first + count > N) return ctx->fail(RGBDR_ERR_OUT_OF_RANGE, "sensor shard outside [0, num_sensors)");
    { int rc_ = system_fence_events(ctx); if (rc_ != RGBDR_OK) return rc_; }  // (the other sensors' frames come from other devices)
    ctx->shard_first = first;
    ctx->shard_count = count;
  }
  if (ctx->shard_first != old_first || ctx->shard_count != old_count) {
    // an upload brings the raw depth (and its pre_morph image) of the shard's layers only: the frame has to be
    // uploaded again before the chain runs on other layers
    ctx->frame_uploaded = false;
    ctx->morph_current = false;
    ctx->shard_pending = false;  // belongs to a frame of the old shard; the next process_textures sets it for the new one
  }
  return RGBDR_OK;
}
RGBDR_CONTAIN(ctx)

int rgbdr_shard_view(rgbdr_ctx* ctx, rgbdr_shard_device_view* out)
try {
  if (!ctx || !out) return RGBDR_ERR_INVALID_ARGUMENT;
  if (!ctx->textures_processed) return ctx->fail(RGBDR_ERR_STATE, "rgbdr_shard_view before process_textures");
  { int rc_ = system_fence_events(ctx); if (rc_ != RGBDR_OK) return rc_; }  // (the host's own collective writes into these buffers)
  const int N = nsens(ctx);
  out->frames = ctx->frame_buf(ctx->rbuf);
  out->sensor_bytes = (size_t)ctx->cfg.depth_w * ctx->cfg.depth_h * sizeof(uint2);
  out->num_sensors = N;
  out->first = ctx->shard_count > 0 ? ctx->shard_first : 0;
  out->count = ctx->shard_count > 0 ? ctx->shard_count : N;
  out->counters = ctx->counters_cur();
  out->num_bricks = (uint32_t)ctx->geo.num_bricks;
  out->stream = (void*)ctx->pstream();
  return RGBDR_OK;  // (a pure getter: the host that runs the collectives itself says so with rgbdr_shard_gather_done)
}
RGBDR_CONTAIN(ctx)

int rgbdr_shard_gather_done(rgbdr_ctx* ctx)
try {
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (!ctx->textures_processed) return ctx->fail(RGBDR_ERR_STATE, "rgbdr_shard_gather_done before process_textures");
  ctx->shard_pending = false;
  return RGBDR_OK;
}
RGBDR_CONTAIN(ctx)

// The frame a context sweeps need not come from its own pre_* chain: rgbdr_import_frame takes the packed frame texels of
// every sensor and the brick counters from device memory (another context's rgbdr_shard_view, after its gather) and leaves
// the context as rgbdr_process_textures would have.  What it is for: a rank whose CHAIN context runs frame k+1 while this
// context still has frame k to sweep, so that the gather of frame k+1 travels under the sweep of frame k
// (rgbd-recon_amd/dist.py LaggedChain).
static int import_pointers(rgbdr_ctx* ctx, const void* packed_frames, const void* brick_counters, hipEvent_t wait_a, hipEvent_t wait_b)
{
  { int rc_ = system_fence_events(ctx); if (rc_ != RGBDR_OK) return rc_; }  // (the frames may come from another device's collective)
  if (ctx->occ_lazy && ctx->occ_lazy_cbuf == ctx->cbuf) {  // the counters a pending filter reads are about to change
    int rc_ = materialise_mask(ctx);
    if (rc_ != RGBDR_OK) return rc_;
  }
  hipStream_t ps = ctx->pstream();
  { int rc_ = join_async_gather(ctx, ps); if (rc_ != RGBDR_OK) return rc_; }
  const int w = ctx->wbuf;
  { int rc_ = wait_last_readers(ctx, w, ps); if (rc_ != RGBDR_OK) return rc_; }  // last readers of buffer w
  if (ctx->imported_rec) HIPCHK(hipStreamWaitEvent(ps, ctx->ev_imported, 0));  // a context still copying out of this one's buffers
  if (wait_a) HIPCHK(hipStreamWaitEvent(ps, wait_a, 0));
  if (wait_b) HIPCHK(hipStreamWaitEvent(ps, wait_b, 0));
  const size_t frame_bytes = (size_t)nsens(ctx) * ctx->cfg.depth_w * ctx->cfg.depth_h * sizeof(uint2);
  HIPCHK(hipMemcpyAsync(ctx->frame_buf(w), packed_frames, frame_bytes, hipMemcpyDeviceToDevice, ps));
  if (brick_counters) {
    // every counter is overwritten: a deferred clearOccupiedBricks has nothing left to do
    HIPCHK(hipMemcpyAsync(ctx->counters_cur(), brick_counters, (size_t)ctx->geo.num_bricks * sizeof(uint32_t), hipMemcpyDeviceToDevice, ps));
    ctx->clear_pending = false;
  }
  ctx->rbuf = w;
  if (ctx->pipelined()) {
    HIPCHK(hipEventRecord(ctx->ev_pre[w], ps));
    ++ctx->pre_serial;
    ctx->ev_pre_rec[w] = true;
  }
  ctx->textures_processed = true;
  ctx->bgmax_for = -1;
  ctx->shard_pending = false;
  ctx->mask_valid = false;  // the occupied filter of the imported counters has not run: rgbdr_update_occupied_bricks
  return RGBDR_OK;
}

int rgbdr_import_frame(rgbdr_ctx* ctx, const void* packed_frames, const void* brick_counters, void* wait_event)
try {
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (!packed_frames) return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, "null frame pointer");
  HIPCHK(hipSetDevice(ctx->device));
  return import_pointers(ctx, packed_frames, brick_counters, (hipEvent_t)wait_event, nullptr);
}
RGBDR_CONTAIN(ctx)

// ... and straight from the context that produced it: behind its chain (whatever stream that ran on) and behind its
// asynchronous gather, if one is under way.  No HIP type crosses the boundary: this is the form a C / C++ host uses.
int rgbdr_import_frame_from(rgbdr_ctx* ctx, rgbdr_ctx* producer)
try {
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (!producer || producer == ctx) return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, "the producer is another context");
  if (!producer->textures_processed) return ctx->fail(RGBDR_ERR_STATE, "rgbdr_import_frame_from: the producer has not processed a frame");
  if (producer->shard_pending) return ctx->fail(RGBDR_ERR_STATE, "rgbdr_import_frame_from: the producer's sensor shard has not been gathered");
  if (producer->device != ctx->device || nsens(producer) != nsens(ctx) || producer->cfg.depth_w != ctx->cfg.depth_w ||
      producer->cfg.depth_h != ctx->cfg.depth_h || producer->geo.num_bricks != ctx->geo.num_bricks ||
      producer->geo.brick_size != ctx->geo.brick_size)
    return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, "producer and consumer differ in device, sensors, image size or brick grid");
  HIPCHK(hipSetDevice(ctx->device));
  if (!producer->ev_export) HIPCHK(hipEventCreateWithFlags(&producer->ev_export, hipEventDisableTiming));
  HIPCHK(hipEventRecord(producer->ev_export, producer->pstream()));  // the producer's chain, in its own stream's order
  int rc = import_pointers(ctx, producer->frame_buf(producer->rbuf), producer->counters_cur(), producer->ev_export,
                           producer->gather_done_rec ? producer->ev_gather_done : nullptr);
  if (rc != RGBDR_OK) return rc;
  // ... and the other way round: the producer's next chain (or gather, or import) overwrites these buffers only behind the
  // copies just enqueued, whatever streams the two contexts run on (a pipelined consumer copies on its second stream)
  if (!producer->ev_imported) HIPCHK(hipEventCreateWithFlags(&producer->ev_imported, hipEventDisableTiming));
  HIPCHK(hipEventRecord(producer->ev_imported, ctx->pstream()));
  producer->imported_rec = true;
  return RGBDR_OK;
}
RGBDR_CONTAIN(ctx)

// the two collectives of the sharded chain on `st`
static int gather_on(rgbdr_ctx* ctx, void* nccl_comm, hipStream_t st, const char* who)
{
  Rccl& r = rccl();
  if (!r.lib || !r.all_gather || !r.all_reduce || !r.comm_count || !r.comm_rank)
    return ctx->fail(RGBDR_ERR_STATE, r.lib ? "the RCCL library lacks ncclAllGather / ncclAllReduce / ncclCommCount" : r.why);
  const int N = nsens(ctx);
  const int first = ctx->shard_count > 0 ? ctx->shard_first : 0, count = ctx->shard_count > 0 ? ctx->shard_count : N;
  auto chk = [&](int rc, const char* what) {
    if (rc == 0) return 0;
    return ctx->fail(RGBDR_ERR_HIP, std::string(what) + ": " + (r.error_string ? r.error_string(rc) : "RCCL error"));
  };
  int world = 0, rank = -1;
  if (chk(r.comm_count(nccl_comm, &world), "ncclCommCount") || chk(r.comm_rank(nccl_comm, &rank), "ncclCommUserRank")) return RGBDR_ERR_HIP;
  // ncclAllGather puts rank r's block at r * count: the shards must be equal and in rank order
  if (count * world != N || first != rank * count)
    return ctx->fail(RGBDR_ERR_STATE, std::string(who) + ": rank r of a k-rank communicator must hold sensors [r n / k, (r + 1) n / k)");
  uint2* frames = ctx->frame_buf(ctx->rbuf);
  const size_t words = (size_t)ctx->cfg.depth_w * ctx->cfg.depth_h * 2 * (size_t)count;  // u32 words of one shard
  tbegin(ctx, "gather", st);
  int rc = chk(r.group_start(), "ncclGroupStart");
  if (rc == 0) rc = chk(r.all_gather((const uint32_t*)frames + words * (size_t)rank, frames, words, /*ncclUint32*/ 3, nccl_comm, st), "ncclAllGather");
  if (rc == 0) rc = chk(r.all_reduce(ctx->counters_cur(), ctx->counters_cur(), (size_t)ctx->geo.num_bricks, /*ncclUint32*/ 3, /*ncclSum*/ 0, nccl_comm, st),
                        "ncclAllReduce");
  const int rc_end = chk(r.group_end(), "ncclGroupEnd");
  tend(ctx, "gather", st);
  return rc != 0 ? rc : rc_end;
}

int rgbdr_shard_allgather(rgbdr_ctx* ctx, void* nccl_comm)
try {
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (!nccl_comm) return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, "null communicator");
  if (!ctx->textures_processed) return ctx->fail(RGBDR_ERR_STATE, "rgbdr_shard_allgather before process_textures");
  HIPCHK(hipSetDevice(ctx->device));
  { int rc_ = join_async_gather(ctx, ctx->pstream()); if (rc_ != RGBDR_OK) return rc_; }
  if (ctx->imported_rec) HIPCHK(hipStreamWaitEvent(ctx->pstream(), ctx->ev_imported, 0));
  const int rc = gather_on(ctx, nccl_comm, ctx->pstream(), "rgbdr_shard_allgather");
  if (rc != RGBDR_OK) return rc;
  ctx->shard_pending = false;
  return RGBDR_OK;
}
RGBDR_CONTAIN(ctx)

// The same collectives OFF the stream the chain ran on: behind an event recorded there, on a stream of the context's own,
// with an event behind them that the consumer of the completed frame waits for -- rgbdr_import_frame_from of the context
// that sweeps it (dist.LaggedChain / host::LaggedChain: the gather of frame k+1 under the sweep of frame k), or whatever
// this context is asked to do with the frame next (every entry point that touches it joins the gather first).
int rgbdr_shard_allgather_async(rgbdr_ctx* ctx, void* nccl_comm)
try {
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (!nccl_comm) return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, "null communicator");
  if (!ctx->textures_processed) return ctx->fail(RGBDR_ERR_STATE, "rgbdr_shard_allgather_async before process_textures");
  HIPCHK(hipSetDevice(ctx->device));
  if (!ctx->gather_stream) {
    HIPCHK(hipStreamCreateWithFlags(&ctx->gather_stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&ctx->ev_gather_from, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&ctx->ev_gather_done, hipEventDisableTiming));
  }
  HIPCHK(hipEventRecord(ctx->ev_gather_from, ctx->pstream()));
  HIPCHK(hipStreamWaitEvent(ctx->gather_stream, ctx->ev_gather_from, 0));
  const int rc = gather_on(ctx, nccl_comm, ctx->gather_stream, "rgbdr_shard_allgather_async");
  if (rc != RGBDR_OK) return rc;
  HIPCHK(hipEventRecord(ctx->ev_gather_done, ctx->gather_stream));
  ctx->gather_done_rec = true;
  ctx->shard_pending = false;
  return RGBDR_OK;
}
RGBDR_CONTAIN(ctx)

int rgbdr_halo_wait(rgbdr_ctx* ctx)
try {
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (ctx->halo_last < 0) return RGBDR_OK;
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_halo_done[ctx->halo_last], 0));
  return RGBDR_OK;
}
RGBDR_CONTAIN(ctx)

}  // extern "C"
