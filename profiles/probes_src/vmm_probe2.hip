// Developer probe: LUT-stream (+ TSDF store) time for arenas mapped with the virtual-memory API
// at a chosen VA alignment (over-reserved, mapped at an aligned offset), next to hipMalloc arenas.
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
  const int K = argc > 1 ? atoi(argv[1]) : 3;
  const size_t lb = (size_t)ntiles * 24576, ob = (size_t)ntiles * 2048;
  v4* out;
  CK(hipMalloc(&out, ob));
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  for (int k = 0; k < K; ++k) {
    v4* m = nullptr;
    CK(hipMalloc(&m, lb));
    printf("hipMalloc            %p  %.3f ms\n", (void*)m, run(m, out));
    for (size_t align : {(size_t)2 << 20, (size_t)64 << 20, (size_t)1 << 30, (size_t)8 << 30}) {
      const size_t total = (lb + ((size_t)2 << 20) - 1) / ((size_t)2 << 20) * ((size_t)2 << 20);
      void* va = nullptr;
      CK(hipMemAddressReserve(&va, total + align, 0, nullptr, 0));
      char* mapped = (char*)(((size_t)va + align - 1) / align * align);
      if (align == ((size_t)2 << 20)) mapped = (char*)va + ((((size_t)va >> 21) & 1) ? 0 : ((size_t)2 << 20));  // odd multiple of 2 MiB
      hipMemGenericAllocationHandle_t h;
      CK(hipMemCreate(&h, total, &prop, 0));
      CK(hipMemMap(mapped, total, 0, h, 0));
      CK(hipMemSetAccess(mapped, total, &acc, 1));
      printf("  vmm align %5zu MiB va %p  %.3f ms\n", align >> 20, (void*)mapped, run((const v4*)mapped, out));
      // keep it mapped: later candidates land elsewhere (leaks until exit, fine for a probe)
    }
  }
  return 0;
}
