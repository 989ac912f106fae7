// Developer probe: does a physically contiguous LUT arena (hipExtMallocWithFlags + hipDeviceMallocContiguous) stream at
// the fast level wherever it lands?  K arenas of the benchmark size from plain hipMalloc (held), then K contiguous ones;
// each is timed with the integrate sweep's memory streams (24 KiB read + 2 KiB written per tile, XCD-chunked order).
//   hipcc -O3 --offload-arch=gfx950 contig_probe.hip -o contig_probe && ./contig_probe [K]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(128) void k_tile(const v4* __restrict__ lut, v4* __restrict__ out, unsigned ntiles, unsigned chunk)
{
  unsigned b = blockIdx.x;
  const unsigned xcd = b & 7u, slot = b >> 3, span = chunk * 8u;
  b = (slot / chunk) * span + xcd * chunk + slot % chunk;
  const v4* q = lut + (size_t)b * 1536;
  v4 acc = {0, 0, 0, 0};
#pragma unroll
  for (int k = 0; k < 12; ++k) acc += __builtin_nontemporal_load(q + k * 128 + threadIdx.x);
  __builtin_nontemporal_store(acc, out + (size_t)b * 128 + threadIdx.x);
}
static const unsigned ntiles = 64 * 64 * 64;
static float run(const v4* a, v4* b)
{
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float t;
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k_tile, dim3(ntiles), dim3(128), 0, 0, a, b, ntiles, 64u);
  hipEventRecord(e0, 0);
  for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(k_tile, dim3(ntiles), dim3(128), 0, 0, a, b, ntiles, 64u);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  hipEventElapsedTime(&t, e0, e1);
  return t / 10;
}
#include <unistd.h>
int main(int argc, char** argv)
{
  const int K = argc > 1 ? atoi(argv[1]) : 6;
  const int mode = argc > 2 ? atoi(argv[2]) : 0;   // 1: contiguous arenas only, in a fresh process, each timed twice
  const size_t lb = (size_t)ntiles * 24576, ob = (size_t)ntiles * 2048;
  v4* out = nullptr;
  if (hipMalloc(&out, ob) != hipSuccess) return 1;
  hipMemset(out, 0, ob);
  std::vector<void*> held;
  if (mode == 1) {
    for (int k = 0; k < K; ++k) {
      void* p = nullptr;
      if (hipExtMallocWithFlags(&p, lb, hipDeviceMallocContiguous) != hipSuccess) { printf("contiguous #%d failed\n", k); break; }
      hipMemset(p, 0, lb);
      hipDeviceSynchronize();
      held.push_back(p);
      printf("contiguous #%d  %p  %.4f ms\n", k, p, run((const v4*)p, out));
    }
    for (size_t k = 0; k < held.size(); ++k) printf("  again      #%zu  %.4f ms\n", k, run((const v4*)held[k], out));
    return 0;
  }
  for (int pass = 0; pass < 2; ++pass) {
    for (int k = 0; k < K; ++k) {
      void* p = nullptr;
      hipError_t e = pass == 0 ? hipMalloc(&p, lb) : hipExtMallocWithFlags(&p, lb, hipDeviceMallocContiguous);
      if (e != hipSuccess) { printf("%s #%d: %s\n", pass ? "contiguous" : "hipMalloc", k, hipGetErrorString(e)); (void)hipGetLastError(); break; }
      hipMemset(p, 0, lb);
      hipDeviceSynchronize();
      printf("%-10s #%d  %p  %.4f ms\n", pass ? "contiguous" : "hipMalloc", k, p, run((const v4*)p, out));
      fflush(stdout);
      held.push_back(p);
    }
    if (pass == 0) {  // release the plain ones so that the contiguous ones can land anywhere
      for (void* p : held) hipFree(p);
      held.clear();
      hipDeviceSynchronize();
      sleep(2);   // released memory is wiped in the background
    }
  }
  // and a contiguous one requested while the first plain allocations are held again (different physical region)
  std::vector<void*> plain;
  for (int k = 0; k < K; ++k) { void* p = nullptr; if (hipMalloc(&p, lb) != hipSuccess) break; hipMemset(p, 0, lb); plain.push_back(p); }
  void* c = nullptr;
  if (hipExtMallocWithFlags(&c, lb, hipDeviceMallocContiguous) == hipSuccess) {
    hipMemset(c, 0, lb);
    hipDeviceSynchronize();
    printf("contiguous behind %zu held plain arenas  %p  %.4f ms\n", plain.size(), c, run((const v4*)c, out));
  } else printf("contiguous behind held arenas: failed\n");
  return 0;
}
