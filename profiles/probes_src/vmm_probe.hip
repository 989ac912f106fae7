// Developer probe: does the LUT stream time depend on how the arena is mapped?  hipMalloc vs the
// virtual-memory API with a chosen VA alignment and physical chunk size.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(128) void k_tile(const v4* __restrict__ lut, v4* __restrict__ out, unsigned ntiles, unsigned chunk)
{
  unsigned b = blockIdx.x;
  if (chunk) {
    const unsigned xcd = b & 7u, slot = b >> 3, span = chunk * 8u;
    b = (slot / chunk) * span + xcd * chunk + slot % chunk;
  }
  const v4* q = lut + (size_t)b * 1536;
  v4 acc = {0, 0, 0, 0};
#pragma unroll
  for (int k = 0; k < 12; ++k) acc += __builtin_nontemporal_load(q + k * 128 + threadIdx.x);
  __builtin_nontemporal_store(acc, out + (size_t)b * 128 + threadIdx.x);
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 0; } } while (0)
static const unsigned ntiles = 64 * 64 * 64;
static float run(const v4* a, v4* b)
{
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float t;
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k_tile, dim3(ntiles), dim3(128), 0, 0, a, b, ntiles, 64u);
  hipEventRecord(e0, 0);
  for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(k_tile, dim3(ntiles), dim3(128), 0, 0, a, b, ntiles, 64u);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  hipEventElapsedTime(&t, e0, e1);
  return t / 10;
}
int main(int argc, char** argv)
{
  const int K = argc > 1 ? atoi(argv[1]) : 4;
  const size_t lb = (size_t)ntiles * 24576, ob = (size_t)ntiles * 2048;
  v4* out;
  CK(hipMalloc(&out, ob));
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  size_t gran = 0, gran_rec = 0;
  CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
  CK(hipMemGetAllocationGranularity(&gran_rec, &prop, hipMemAllocationGranularityRecommended));
  printf("granularity min %zu recommended %zu\n", gran, gran_rec);
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  for (int k = 0; k < K; ++k) {
    v4* m = nullptr;
    CK(hipMalloc(&m, lb));
    hipMemset(m, 0, lb);
    printf("hipMalloc      %p  %.3f ms\n", (void*)m, run(m, out));
    // keep m alive so the VMM arenas land elsewhere
    for (size_t chunk : {(size_t)0, (size_t)1 << 30, (size_t)256 << 20, (size_t)32 << 20}) {
      const size_t align = (size_t)1 << 30;
      const size_t total = (lb + align - 1) / align * align;
      void* va = nullptr;
      CK(hipMemAddressReserve(&va, total, align, nullptr, 0));
      std::vector<hipMemGenericAllocationHandle_t> hs;
      const size_t c = chunk ? chunk : total;
      for (size_t off = 0; off < total; off += c) {
        hipMemGenericAllocationHandle_t h;
        CK(hipMemCreate(&h, c, &prop, 0));
        CK(hipMemMap((char*)va + off, c, 0, h, 0));
        hs.push_back(h);
      }
      CK(hipMemSetAccess(va, total, &acc, 1));
      hipMemset(va, 0, lb);
      printf("  vmm chunk %5zu MiB va %p  %.3f ms\n", c >> 20, va, run((const v4*)va, out));
      CK(hipMemUnmap(va, total));
      for (auto h : hs) CK(hipMemRelease(h));
      CK(hipMemAddressFree(va, total));
    }
  }
  return 0;
}
