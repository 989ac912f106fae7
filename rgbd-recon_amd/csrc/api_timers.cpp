// api_timers.cpp -- TimerDatabase (framework/rendering/timer_database.cpp:26-121) on HIP events.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>

#include "context.hpp"

using namespace rgbdr;

namespace rgbdr {
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

// Accumulating mode keeps an event pair per interval until rgbdr_timer_stats reads them.  Past kMaxPending the oldest half
// is read here (those intervals ended long ago: the wait is for the first one at most) and kept as a sum.
constexpr size_t kMaxPending = 2048;
static void fold_oldest(Timer& t)
{
  const size_t n = t.pending.size() / 2;
  for (size_t i = 0; i < n; ++i) {
    const auto ev = t.pending[i];
    float ms = 0.0f;
    if (ev.second && hipEventSynchronize(ev.second) == hipSuccess && hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess) {
      t.folded_ns += (double)ms * 1.0e6;
      ++t.folded_count;
    } else {
      (void)hipGetLastError();  // begin without end: dropped, as rgbdr_timer_stats does
    }
    t.pool.push_back(ev);
  }
  t.pending.erase(t.pending.begin(), t.pending.begin() + (std::ptrdiff_t)n);
}

void tbegin(rgbdr_ctx* c, const char* name, hipStream_t st)
{
  if (timer_muted(c, name)) return;
  Timer& t = c->tm[name];
  if (c->accumulate) {
    if (t.pending.size() >= kMaxPending) fold_oldest(t);
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
void tend(rgbdr_ctx* c, const char* name, hipStream_t st)
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

void destroy_timers(rgbdr_ctx* ctx)
{
  for (auto& kv : ctx->tm) {
    if (kv.second.a) (void)hipEventDestroy(kv.second.a);
    if (kv.second.b) (void)hipEventDestroy(kv.second.b);
    for (auto* v : {&kv.second.pending, &kv.second.pool})
      for (auto& ev : *v) {
        (void)hipEventDestroy(ev.first);
        (void)hipEventDestroy(ev.second);
      }
  }
  ctx->tm.clear();
}
}  // namespace rgbdr

extern "C" {

int rgbdr_set_timer_detail(rgbdr_ctx* ctx, int detail)
try {
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  ctx->timer_detail = detail < 1 ? 0 : (detail < 2 ? 1 : 2);
  return RGBDR_OK;
}
RGBDR_CONTAIN(ctx)

int rgbdr_enable_timers(rgbdr_ctx* ctx, int on)
try {
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  ctx->timers = on != 0;
  return RGBDR_OK;
}
RGBDR_CONTAIN(ctx)

int rgbdr_enable_timer_accumulation(rgbdr_ctx* ctx, int on)
try {
  if (!ctx) return RGBDR_ERR_INVALID_ARGUMENT;
  ctx->accumulate = on != 0;
  if (on) ctx->timers = true;
  return RGBDR_OK;
}
RGBDR_CONTAIN(ctx)

int rgbdr_timer_stats(rgbdr_ctx* ctx, const char* name, uint64_t* total_ns, uint32_t* count)
try {
  if (!ctx || !name || !total_ns || !count) return RGBDR_ERR_INVALID_ARGUMENT;
  *total_ns = 0;
  *count = 0;
  auto it = ctx->tm.find(name);
  if (it == ctx->tm.end()) return ctx->fail(RGBDR_ERR_OUT_OF_RANGE, std::string("no timer ") + name);
  // the intervals may have been recorded on either stream (pipelined mode: the pre_* timers
  // and "bricks" live on the second one)
  { int rc_ = sync_all(ctx); if (rc_ != RGBDR_OK) return rc_; }
  double total = it->second.folded_ns;
  *count = it->second.folded_count;
  it->second.folded_ns = 0.0;
  it->second.folded_count = 0;
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
RGBDR_CONTAIN(ctx)

int rgbdr_timer_ns(rgbdr_ctx* ctx, const char* name, uint64_t* ns)
try {
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
RGBDR_CONTAIN(ctx)

}  // extern "C"
