// Developer probe: what is slow about a slow placement?  Three physically contiguous 6.4 GB arenas in a fresh process (the
// first is usually in the slow zone, the third in the fast one); for each: the sweep's read stream with non-temporal and
// with plain loads, read-only (no stores), and one XCD at a time (blocks of the other seven exit at once).
//   hipcc -O3 --offload-arch=gfx950 zone_probe.hip -o zone_probe && ./zone_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v4 __attribute__((ext_vector_type(4)));
template <bool NT, int STORE>
__global__ __launch_bounds__(128) void k_tile(const v4* __restrict__ lut, v4* __restrict__ out, unsigned ntiles, unsigned chunk, int only_xcd)
{
  unsigned b = blockIdx.x;
  const unsigned xcd = b & 7u, slot = b >> 3, span = chunk * 8u;
  if (only_xcd >= 0 && (int)xcd != only_xcd) return;
  b = (slot / chunk) * span + xcd * chunk + slot % chunk;
  const v4* q = lut + (size_t)b * 1536;
  v4 acc = {0, 0, 0, 0};
#pragma unroll
  for (int k = 0; k < 12; ++k) acc += NT ? __builtin_nontemporal_load(q + k * 128 + threadIdx.x) : q[k * 128 + threadIdx.x];
  if (STORE == 1) __builtin_nontemporal_store(acc, out + (size_t)b * 128 + threadIdx.x);
  else if (STORE == 2) out[(size_t)b * 128 + threadIdx.x] = acc;
  else if (acc.x == 12345.678f) out[0] = acc;
}
// T consecutive tiles per block, all loads first... then the T output tiles stored back to back (T * 2 KiB contiguous)
template <int T>
__global__ __launch_bounds__(128) void k_multi(const v4* __restrict__ lut, v4* __restrict__ out, unsigned ngroups, unsigned chunk)
{
  unsigned b = blockIdx.x;
  const unsigned xcd = b & 7u, slot = b >> 3, span = chunk * 8u;
  b = (slot / chunk) * span + xcd * chunk + slot % chunk;
  if (b >= ngroups) return;
  v4 acc[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const v4* q = lut + ((size_t)b * T + t) * 1536;
    acc[t] = v4{0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[t] += __builtin_nontemporal_load(q + k * 128 + threadIdx.x);
  }
#pragma unroll
  for (int t = 0; t < T; ++t) __builtin_nontemporal_store(acc[t], out + ((size_t)b * T + t) * 128 + threadIdx.x);
}
static const unsigned ntiles = 64 * 64 * 64;
template <int T>
static float run_multi(const v4* a, v4* b)
{
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float t;
  const unsigned ng = ntiles / T;
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_multi<T>), dim3(ng), dim3(128), 0, 0, a, b, ng, 64u / T);
  hipEventRecord(e0, 0);
  for (int r = 0; r < 6; ++r) hipLaunchKernelGGL((k_multi<T>), dim3(ng), dim3(128), 0, 0, a, b, ng, 64u / T);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  hipEventElapsedTime(&t, e0, e1);
  return t / 6;
}
template <bool NT, int STORE>
static float run(const v4* a, v4* b, int only_xcd = -1)
{
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float t;
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_tile<NT, STORE>), dim3(ntiles), dim3(128), 0, 0, a, b, ntiles, 64u, only_xcd);
  hipEventRecord(e0, 0);
  for (int r = 0; r < 6; ++r) hipLaunchKernelGGL((k_tile<NT, STORE>), dim3(ntiles), dim3(128), 0, 0, a, b, ntiles, 64u, only_xcd);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  hipEventElapsedTime(&t, e0, e1);
  return t / 6;
}
int main(int argc, char** argv)
{
  const int NA = argc > 1 ? atoi(argv[1]) : 4, NO = argc > 2 ? atoi(argv[2]) : 6;
  const size_t lb = (size_t)ntiles * 24576, ob = (size_t)ntiles * 2048;
  std::vector<v4*> outs;
  for (int k = 0; k < NO; ++k) {
    v4* o = nullptr;
    if (hipMalloc(&o, ob) != hipSuccess) break;
    hipMemset(o, 0, ob);
    outs.push_back(o);
  }
  // one more volume with 8 offsets inside it (is the pairing a matter of the relative phase of the two streams?)
  v4* big = nullptr;
  hipMalloc(&big, ob + (8u << 20));
  hipMemset(big, 0, ob + (8u << 20));
  for (int k = 0; k < NA; ++k) {
    void* p = nullptr;
    if (hipMalloc(&p, lb) != hipSuccess) { printf("alloc %d failed\n", k); break; }
    hipMemset(p, 0, lb);
    hipDeviceSynchronize();
    const v4* a = (const v4*)p;
    printf("arena #%d %p: read-only %.4f | with volume j:", k, p, run<true, 0>(a, outs[0]));
    for (size_t j = 0; j < 2 && j < outs.size(); ++j) printf(" %.4f", run<true, 1>(a, outs[j]));
    printf(" | tiles per block 1/2/4/8: %.4f %.4f %.4f %.4f", run_multi<1>(a, outs[0]), run_multi<2>(a, outs[0]), run_multi<4>(a, outs[0]), run_multi<8>(a, outs[0]));
    printf(" | plain stores: %.4f  plain loads + plain stores: %.4f  plain loads + nt stores: %.4f", run<true, 2>(a, outs[0]), run<false, 2>(a, outs[0]), run<false, 1>(a, outs[0]));
    printf("\n");
    fflush(stdout);
  }
  return 0;
}
