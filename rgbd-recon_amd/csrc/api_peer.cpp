// api_peer.cpp -- the Z-slab halo moved by the copy engine: within one node a face is one
// contiguous range, so instead of RCCL's send / recv kernels -- which compete with the sweep for wavefront slots and are
// enqueued by a host call that has stalled for tens of milliseconds -- each rank PULLS its neighbours' staged faces with
// hipMemcpyAsync from their staging sets, mapped once through HIP IPC.  Ordering across the processes is by STEP NUMBERS in
// a small block of shared memory that every process registers with HIP (hipHostRegister) and streams write / wait for
// (hipStreamWriteValue32 / hipStreamWaitValue32, >=):
//   * every rank owns a block of INBOX words its own streams wait on and its neighbours' streams write: behind the sweep
//     that filled staging set b a rank writes `step` to its neighbours' staged_in[.][b]; a rank's side stream waits for its
//     own staged_in[side][b] >= step, copies that neighbour's face, and writes `step` to the neighbour's pulled_in[.][b],
//     which the neighbour's stream waits for (>= step - 2) before the sweep of two steps later refills the set.
//   * a rank waits on its OWN words only, so it can always release its streams itself (destroy, a neighbour given up on).
//   * the numbers only grow, so a wait cannot latch on to an older record (the trap of interprocess EVENTS, whose wait acts
//     on the most recent record at the time of the call -- and whose ring of 32 signals a host that enqueues forty steps
//     ahead overruns: "invalid argument" after a few dozen steps).
//   * the host looks at the same words before it enqueues a wait: a neighbour more than kMaxLag steps behind is waited for
//     (back-pressure on a host running ahead) and reported as gone after RGBDR_PEER_TIMEOUT_S instead of hanging the stream.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include "context.hpp"

using namespace rgbdr;

namespace rgbdr {
// the words both sides of an exchange watch; lives in POSIX shared memory owned by the exporting context
struct PeerSeq {
  uint32_t staged_in[2][2];  // [side][set]: the neighbour on `side` has the faces of step (1, 2, ...) in its staging set b
  uint32_t pulled_in[2][2];  // [side][set]: the neighbour on `side` has copied this rank's face of that step out of set b
  uint32_t pad[1024 - 8];    // (one page: hipHostRegister works on pages)
};
static_assert(sizeof(PeerSeq) == 4096, "one page");
constexpr uint32_t kMaxLag = 24;  // steps a host may enqueue ahead of what a neighbour's device has finished

struct HaloWire {  // what an rgbdr_halo_peer carries (opaque to the host)
  uint32_t magic, version;
  int32_t pid, device;
  uint64_t face_bytes;
  hipIpcMemHandle_t mem[2][2];        // [set][face 0 = lowest owned layers, 1 = highest]
  uint64_t raw_mem[2][2], raw_seq;    // the same objects for a peer in the owner's own process
  char shm_name[48];
};
static_assert(sizeof(HaloWire) <= sizeof(rgbdr_halo_peer), "rgbdr_halo_peer too small");
constexpr uint32_t kWireMagic = 0x52474244u;  // "RGBD"

struct PeerLink {  // one neighbour as this context sees it
  bool set = false, same_process = false;
  void* mem[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
  PeerSeq* seq = nullptr;  // the neighbour's words, registered with HIP in this process
};

struct PeerState {
  bool exported = false;
  unsigned base_step = 0;  // the context's halo_step when it was (last) exported: the protocol counts steps from there
  PeerSeq* seq = nullptr;  // this context's words
  // On this ROCm a stream's write / wait-for-value is a one-wavefront kernel of 4-6 us.  None of them runs on the context's
  // stream: the waits for "the neighbours have copied set b" sit on a stream of their own and reach the sweep as an event.
  hipStream_t wait_stream = nullptr;
  hipEvent_t ev_refill[2] = {nullptr, nullptr};
  char shm_name[48] = {0};
  PeerLink link[2];  // 0: lower neighbour, 1: upper
};

static void unmap_link(PeerLink& l)
{
  if (!l.set) return;
  if (!l.same_process) {
    for (int b = 0; b < 2; ++b)
      for (int f = 0; f < 2; ++f)
        if (l.mem[b][f]) (void)hipIpcCloseMemHandle(l.mem[b][f]);
    if (l.seq) {
      (void)hipHostUnregister(l.seq);
      (void)munmap(l.seq, sizeof(PeerSeq));
    }
  }
  l = PeerLink{};
}

// every wait this context's streams may still sit in is on a word of its own block: let them through (a neighbour that died,
// a context destroyed mid-step) -- the copies behind them then move whatever the neighbour's staging sets hold
void release_peer_waits(rgbdr_ctx* ctx, int side)
{
  PeerState* p = ctx->peer;
  if (!p || !p->seq) return;
  for (int s = 0; s < 2; ++s) {
    if (side >= 0 && s != side) continue;
    for (int b = 0; b < 2; ++b) {
      volatile uint32_t* w[2] = {&p->seq->staged_in[s][b], &p->seq->pulled_in[s][b]};
      for (volatile uint32_t* x : w) *x = *x + 0x40000000u;  // (>= every step a queued wait can name)
    }
  }
}

void destroy_peer_state(rgbdr_ctx* ctx)
{
  PeerState* p = ctx->peer;
  if (!p) return;
  release_peer_waits(ctx, -1);
  unmap_link(p->link[0]);
  unmap_link(p->link[1]);
  if (p->wait_stream) {
    (void)hipStreamSynchronize(p->wait_stream);
    (void)hipStreamDestroy(p->wait_stream);
  }
  for (hipEvent_t e : p->ev_refill)
    if (e) (void)hipEventDestroy(e);
  if (p->seq) {
    (void)hipHostUnregister(p->seq);
    (void)munmap(p->seq, sizeof(PeerSeq));
  }
  if (p->shm_name[0]) (void)shm_unlink(p->shm_name);
  delete p;
  ctx->peer = nullptr;
}
}  // namespace rgbdr

namespace {
double peer_timeout_s()
{
  const char* e = std::getenv("RGBDR_PEER_TIMEOUT_S");
  const double v = e ? std::atof(e) : 30.0;
  return v > 0.0 ? v : 30.0;
}

// the neighbour's device has written `step` (or a later one) to the word
bool wait_seq(const uint32_t& word, uint32_t step)
{
  const volatile uint32_t& w = word;
  if ((int32_t)(w - step) >= 0) return true;
  const auto t_end = std::chrono::steady_clock::now() + std::chrono::duration<double>(peer_timeout_s());
  for (int spin = 0;; ++spin) {
    if ((int32_t)(w - step) >= 0) return true;
    if (spin > 2000) {
      if (std::chrono::steady_clock::now() > t_end) return false;
      std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
  }
}
}  // namespace

extern "C" {
int rgbdr_halo_export(rgbdr_ctx* ctx, rgbdr_halo_peer* out)
try {
  if (!ctx || !out) return RGBDR_ERR_INVALID_ARGUMENT;
  if (ctx->halo <= 0) return ctx->fail(RGBDR_ERR_STATE, "rgbdr_halo_export needs a Z-slab context (slab_count > 1)");
  HIPCHK(hipSetDevice(ctx->device));
  for (int b = 0; b < 2; ++b) {
    int rc_ = rgbdr_halo_staging(ctx, b, nullptr, nullptr, nullptr);
    if (rc_ != RGBDR_OK) return rc_;
  }
  if (!ctx->peer) ctx->peer = new PeerState();
  PeerState& P = *ctx->peer;
  if (!P.exported) {
    std::snprintf(P.shm_name, sizeof P.shm_name, "/rgbdr_halo_%d_%llx", (int)getpid(), (unsigned long long)(uintptr_t)ctx & 0xffffffffffULL);
    (void)shm_unlink(P.shm_name);
    const int fd = shm_open(P.shm_name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, sizeof(PeerSeq)) != 0) {
      if (fd >= 0) close(fd);
      P.shm_name[0] = 0;
      return ctx->fail(RGBDR_ERR_IO, "rgbdr_halo_export: cannot create the shared step block in /dev/shm");
    }
    void* m = mmap(nullptr, sizeof(PeerSeq), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return ctx->fail(RGBDR_ERR_IO, "rgbdr_halo_export: mmap of the shared step block failed");
    std::memset(m, 0, sizeof(PeerSeq));
    if (hipHostRegister(m, sizeof(PeerSeq), hipHostRegisterDefault) != hipSuccess) {
      (void)hipGetLastError();
      (void)munmap(m, sizeof(PeerSeq));
      return ctx->fail(RGBDR_ERR_HIP, "rgbdr_halo_export: hipHostRegister of the shared step block failed");
    }
    P.seq = (PeerSeq*)m;
    HIPCHK(hipStreamCreateWithFlags(&P.wait_stream, hipStreamNonBlocking));
    for (int b = 0; b < 2; ++b) HIPCHK(hipEventCreateWithFlags(&P.ev_refill[b], hipEventDisableTiming));
    P.exported = true;
  }
  // (re-)exporting starts the protocol over: every rank exports at the same step of its loop, then sets its peers
  { int rc_ = sync_all(ctx); if (rc_ != RGBDR_OK) return rc_; }
  HIPCHK(hipStreamSynchronize(P.wait_stream));
  P.base_step = ctx->halo_step;
  std::memset(P.seq, 0, 8 * sizeof(uint32_t));
  HaloWire w;
  std::memset(&w, 0, sizeof w);
  w.magic = kWireMagic;
  w.version = 2;
  w.pid = (int32_t)getpid();
  w.device = ctx->device;
  w.face_bytes = ctx->layer_floats * (size_t)ctx->halo * sizeof(float);
  for (int b = 0; b < 2; ++b)
    for (int f = 0; f < 2; ++f) {
      HIPCHK(hipIpcGetMemHandle(&w.mem[b][f], ctx->d_stage[b][f]));
      w.raw_mem[b][f] = (uint64_t)(uintptr_t)ctx->d_stage[b][f];
    }
  w.raw_seq = (uint64_t)(uintptr_t)P.seq;
  std::memcpy(w.shm_name, P.shm_name, sizeof w.shm_name);
  std::memset(out, 0, sizeof *out);
  std::memcpy(out, &w, sizeof w);
  return RGBDR_OK;
}
RGBDR_CONTAIN(ctx)

int rgbdr_halo_set_peer(rgbdr_ctx* ctx, int side, const rgbdr_halo_peer* peer)
try {
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (side < 0 || side > 1) return ctx->fail(RGBDR_ERR_OUT_OF_RANGE, "rgbdr_halo_set_peer: side is 0 (lower neighbour) or 1 (upper)");
  if (ctx->halo <= 0) return ctx->fail(RGBDR_ERR_STATE, "rgbdr_halo_set_peer needs a Z-slab context (slab_count > 1)");
  if (!ctx->peer || !ctx->peer->exported) return ctx->fail(RGBDR_ERR_STATE, "rgbdr_halo_set_peer before rgbdr_halo_export of this context");
  if (peer && (side == 0 ? ctx->cfg.slab_rank <= 0 : ctx->cfg.slab_rank >= ctx->cfg.slab_count - 1))
    return ctx->fail(RGBDR_ERR_STATE, "rgbdr_halo_set_peer: this slab has no neighbour on that side (the first slab has none below, the last none above)");
  HIPCHK(hipSetDevice(ctx->device));
  PeerLink& l = ctx->peer->link[side];
  if (!peer) release_peer_waits(ctx, side);  // the neighbour is gone: nothing queued may wait for it any longer
  { int rc_ = sync_all(ctx); if (rc_ != RGBDR_OK) return rc_; }  // (queued waits and copies still read the old mapping)
  HIPCHK(hipStreamSynchronize(ctx->peer->wait_stream));
  unmap_link(l);
  if (!peer) {
    std::memset(ctx->peer->seq->staged_in[side], 0, sizeof ctx->peer->seq->staged_in[side]);
    std::memset(ctx->peer->seq->pulled_in[side], 0, sizeof ctx->peer->seq->pulled_in[side]);
    return RGBDR_OK;
  }
  HaloWire w;
  std::memcpy(&w, peer, sizeof w);
  if (w.magic != kWireMagic || w.version != 2) return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, "rgbdr_halo_set_peer: not an rgbdr_halo_peer of this library version");
  if (w.face_bytes != ctx->layer_floats * (size_t)ctx->halo * sizeof(float))
    return ctx->fail(RGBDR_ERR_INVALID_ARGUMENT, "rgbdr_halo_set_peer: the neighbour's faces have another size (same grid x / y and halo on every slab)");
  l.same_process = w.pid == (int32_t)getpid();
  if (l.same_process) {  // its objects as they are
    l.seq = (PeerSeq*)(uintptr_t)w.raw_seq;
    for (int b = 0; b < 2; ++b)
      for (int f = 0; f < 2; ++f) l.mem[b][f] = (void*)(uintptr_t)w.raw_mem[b][f];
    l.set = true;
    return RGBDR_OK;
  }
  w.shm_name[sizeof w.shm_name - 1] = 0;
  const int fd = shm_open(w.shm_name, O_RDWR, 0600);
  if (fd < 0) return ctx->fail(RGBDR_ERR_IO, std::string("rgbdr_halo_set_peer: cannot open the neighbour's step block ") + w.shm_name + " (same node only)");
  void* m = mmap(nullptr, sizeof(PeerSeq), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (m == MAP_FAILED) return ctx->fail(RGBDR_ERR_IO, "rgbdr_halo_set_peer: mmap of the neighbour's step block failed");
  if (hipHostRegister(m, sizeof(PeerSeq), hipHostRegisterDefault) != hipSuccess) {
    (void)hipGetLastError();
    (void)munmap(m, sizeof(PeerSeq));
    return ctx->fail(RGBDR_ERR_HIP, "rgbdr_halo_set_peer: hipHostRegister of the neighbour's step block failed");
  }
  l.seq = (PeerSeq*)m;
  l.set = true;  // (from here on unmap_link releases what has been opened)
  for (int b = 0; b < 2; ++b)
    for (int f = 0; f < 2; ++f) HIPCHK(hipIpcOpenMemHandle(&l.mem[b][f], w.mem[b][f], hipIpcMemLazyEnablePeerAccess));
  return RGBDR_OK;
}
RGBDR_CONTAIN(ctx)
}  // extern "C"

namespace {
// `st` goes on once *word >= step.  The host looks first: a neighbour whose DEVICE is more than kMaxLag steps behind what is
// being enqueued is waited for here (a host running ahead is held back), and given up on after the time-out.
int wait_word(rgbdr_ctx* ctx, hipStream_t st, uint32_t* word, uint32_t step, const char* what)
{
  if (step > kMaxLag && !wait_seq(*word, step - kMaxLag))
    return ctx->fail(RGBDR_ERR_STATE, std::string("halo (copy engine): the neighbour ") + what + " (dead, or not stepping with this rank)");
  HIPCHK(hipStreamWaitValue32(st, word, step, hipStreamWaitValueGte, 0xffffffffu));
  return RGBDR_OK;
}
}  // namespace

// the peer form of rgbdr_halo_begin_step's wait: the neighbours' copies out of staging set b of two steps ago
int rgbdr::peer_begin_step(rgbdr_ctx* ctx, int b, uint32_t absolute_step)
{
  PeerState* P = ctx->peer;
  if (!P || !P->exported || absolute_step <= P->base_step + 2) return RGBDR_OK;
  const uint32_t step = absolute_step - P->base_step;
  bool any = false;
  for (int side = 0; side < 2; ++side) {
    PeerLink& l = P->link[side];
    if (!l.set) continue;
    int rc_ = wait_word(ctx, P->wait_stream, &P->seq->pulled_in[side][b], step - 2, "has stopped pulling this rank's faces");
    if (rc_ != RGBDR_OK) return rc_;
    any = true;
  }
  if (any) {
    HIPCHK(hipEventRecord(P->ev_refill[b], P->wait_stream));
    HIPCHK(hipStreamWaitEvent(ctx->stream, P->ev_refill[b], 0));
  }
  return RGBDR_OK;
}

extern "C" {
int rgbdr_halo_pull_async(rgbdr_ctx* ctx)
try {
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  if (!ctx->peer || !ctx->peer->exported) return ctx->fail(RGBDR_ERR_STATE, "rgbdr_halo_pull_async before rgbdr_halo_export / rgbdr_halo_set_peer");
  if (!ctx->halo_begun) return ctx->fail(RGBDR_ERR_STATE, "rgbdr_halo_pull_async before rgbdr_halo_begin_step + rgbdr_integrate");
  HIPCHK(hipSetDevice(ctx->device));
  PeerState& P = *ctx->peer;
  const int b = ctx->halo_step & 1;
  if (ctx->stage_target != b || !ctx->halo_staged)
    return ctx->fail(RGBDR_ERR_STATE, "rgbdr_halo_pull_async: no rgbdr_integrate has filled the staging set since rgbdr_halo_begin_step");
  const uint32_t step = ctx->halo_step - P.base_step + 1;
  ctx->halo_begun = false;
  ++ctx->halo_step;
  // my side stream: behind my own staging (hence behind every earlier reader of my halo layers on the context's stream)
  HIPCHK(hipEventRecord(ctx->ev_halo_staged[b], ctx->stream));
  HIPCHK(hipStreamWaitEvent(ctx->halo_stream, ctx->ev_halo_staged[b], 0));
  // my faces of this step are staged: the step number into the neighbours' inboxes (I am the neighbour on their OTHER side)
  for (int side = 0; side < 2; ++side)
    if (P.link[side].set) HIPCHK(hipStreamWriteValue32(ctx->halo_stream, &P.link[side].seq->staged_in[1 - side][b], step, 0));
  const size_t face = ctx->layer_floats * (size_t)ctx->halo;
  const int owned = ctx->geo.slab_tile_z1 - ctx->geo.slab_tile_z0;
  float* recv[2] = {ctx->d_tsdf_base, ctx->d_tsdf_owned + ctx->layer_floats * (size_t)owned};
  tbegin(ctx, "halo", ctx->halo_stream);
  for (int side = 0; side < 2; ++side) {
    PeerLink& l = P.link[side];
    if (!l.set) continue;
    int rc_ = wait_word(ctx, ctx->halo_stream, &P.seq->staged_in[side][b], step, "has stopped staging its faces");
    if (rc_ != RGBDR_OK) return rc_;
    // the lower neighbour's HIGHEST layers border my lowest, and the other way round
    HIPCHK(hipMemcpyAsync(recv[side], l.mem[b][1 - side], face * sizeof(float), hipMemcpyDeviceToDevice, ctx->halo_stream));
    HIPCHK(hipStreamWriteValue32(ctx->halo_stream, &l.seq->pulled_in[1 - side][b], step, 0));
  }
  tend(ctx, "halo", ctx->halo_stream);
  HIPCHK(hipEventRecord(ctx->ev_halo_done[b], ctx->halo_stream));
  ctx->halo_done_rec[b] = true;
  ctx->halo_last = b;
  return RGBDR_OK;
}
RGBDR_CONTAIN(ctx)
}  // extern "C"
