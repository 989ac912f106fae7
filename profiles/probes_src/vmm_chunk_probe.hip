// Does the "slow zone" of the device memory (a LUT arena streams 11 % slower next to the TSDF store stream
// when hipMalloc places it badly) show per 1-GiB chunk of physical memory, and can chunks be picked?  Creates K physical
// chunks with the virtual-memory API (hipMemCreate), maps them into one reserved range, and times the sweep's pair of
// streams (24 KB read : 2 KB written per block) on every chunk; then the same per 1-GiB piece of one plain hipMalloc.
//   hipcc --offload-arch=gfx950 -O3 -o vmm_chunk_probe vmm_chunk_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));
// one 128-thread block per "tile": 12 planes of 2 KB read (16 B per lane each), 2 KB stored
__global__ __launch_bounds__(128) void k_pair(const v4f* __restrict__ src, v4f* __restrict__ sink, unsigned tiles)
{
  const unsigned t = blockIdx.x;
  const v4f* p = src + (size_t)t * 12 * 128 + threadIdx.x;
  v4f acc = {0, 0, 0, 0};
#pragma unroll
  for (int k = 0; k < 12; ++k) acc += __builtin_nontemporal_load(p + k * 128);
  __builtin_nontemporal_store(acc, sink + (size_t)t * 128 + threadIdx.x);
}

// one 16-byte load per 4 KiB of the range (a test of translation, not of bandwidth), every lane of a wavefront on its own page
__global__ __launch_bounds__(256) void k_touch(const v4f* __restrict__ src, v4f* __restrict__ sink, unsigned n, unsigned stride16)
{
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // pages in a scattered order (a multiplicative hash modulo n, n a power of two) so that neighbouring lanes do not share a 2-MiB page
  const unsigned j = (i * 2654435761u) & (n - 1);
  v4f a = __builtin_nontemporal_load(src + (size_t)j * stride16);
  if (a.x == 123.456f) sink[i] = a;
}
static float time_touch(const void* base, size_t bytes, void* sink, hipEvent_t e0, hipEvent_t e1)
{
  const unsigned n = (unsigned)(bytes / 4096);
  float best = 1e9f;
  for (int r = 0; r < 4; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_touch, dim3((n + 255) / 256), dim3(256), 0, 0, (const v4f*)base, (v4f*)sink, n, 256u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (r > 0 && ms < best) best = ms;
  }
  return best;
}

static float time_range(const void* base, size_t bytes, void* sink, hipEvent_t e0, hipEvent_t e1)
{
  const unsigned tiles = (unsigned)(bytes / (12 * 2048));
  float best = 1e9f;
  for (int r = 0; r < 4; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_pair, dim3(tiles), dim3(128), 0, 0, (const v4f*)base, (v4f*)sink, tiles);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (r > 0 && ms < best) best = ms;
  }
  return best;
}

int main(int argc, char** argv)
{
  const int K = argc > 1 ? atoi(argv[1]) : 40;
  const size_t chunk = (size_t)1 << 30;
  CHK(hipSetDevice(0));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  void* sink;
  CHK(hipMalloc(&sink, 13 * chunk / 12 + ((size_t)1 << 20)));   // (the widest range timed below is 13 chunks)
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  size_t gran = 0;
  CHK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
  printf("allocation granularity %zu bytes\n", gran); fflush(stdout); setvbuf(stdout, nullptr, _IONBF, 0);
  void* va = nullptr;
  CHK(hipMemAddressReserve(&va, (size_t)K * chunk, 0, nullptr, 0));
  std::vector<hipMemGenericAllocationHandle_t> h(K);
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  int made = 0;
  for (int i = 0; i < K; ++i) {
    if (hipMemCreate(&h[i], chunk, &prop, 0) != hipSuccess) { (void)hipGetLastError(); break; }
    CHK(hipMemMap((char*)va + (size_t)i * chunk, chunk, 0, h[i], 0));
    ++made;
  }
  CHK(hipMemSetAccess(va, (size_t)made * chunk, &acc, 1));
  CHK(hipMemset(va, 0, (size_t)made * chunk));
  CHK(hipDeviceSynchronize());
  const double gb = (double)chunk * (13.0 / 12.0) / 1e9;
  printf("%d chunks of 1 GiB created with hipMemCreate; GB/s of the stream pair per chunk:\n", made);
  std::vector<float> rate(made);
  for (int i = 0; i < made; ++i) {
    rate[i] = (float)(gb / (time_range((char*)va + (size_t)i * chunk, chunk, sink, e0, e1) * 1e-3));
    printf("%5.0f%s", rate[i], (i % 10 == 9 || i == made - 1) ? "\n" : " ");
  }
  // translation: one load per 4 KiB, scattered, on the three fastest and the three slowest chunks
  {
    std::vector<int> o(made);
    for (int i = 0; i < made; ++i) o[i] = i;
    std::sort(o.begin(), o.end(), [&](int a, int b) { return rate[a] > rate[b]; });
    printf("one scattered 16-byte load per 4 KiB of a chunk (262144 loads), us:  fastest chunks");
    for (int k = 0; k < 3 && k < made; ++k) printf(" %.1f", 1e3f * time_touch((char*)va + (size_t)o[k] * chunk, chunk, sink, e0, e1));
    printf("   slowest chunks");
    for (int k = 0; k < 3 && k < made; ++k) printf(" %.1f", 1e3f * time_touch((char*)va + (size_t)o[made - 1 - k] * chunk, chunk, sink, e0, e1));
    printf("\n");
  }
  // the whole mapped range at once, and the best 13 chunks remapped contiguously
  if (made >= 13) {
    printf("first 13 chunks as one range: %.0f GB/s\n", 13 * gb / (time_range(va, 13 * chunk, sink, e0, e1) * 1e-3));
    std::vector<int> order(made);
    for (int i = 0; i < made; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return rate[a] > rate[b]; });
    void* va2 = nullptr;
    CHK(hipMemAddressReserve(&va2, 13 * chunk, 0, nullptr, 0));
    for (int i = 0; i < 13; ++i) CHK(hipMemMap((char*)va2 + (size_t)i * chunk, chunk, 0, h[order[i]], 0));
    CHK(hipMemSetAccess(va2, 13 * chunk, &acc, 1));
    printf("the 13 fastest chunks mapped as one range: %.0f GB/s;", 13 * gb / (time_range(va2, 13 * chunk, sink, e0, e1) * 1e-3));
    void* va3 = nullptr;
    CHK(hipMemAddressReserve(&va3, 13 * chunk, 0, nullptr, 0));
    for (int i = 0; i < 13; ++i) CHK(hipMemMap((char*)va3 + (size_t)i * chunk, chunk, 0, h[order[made - 1 - i]], 0));
    CHK(hipMemSetAccess(va3, 13 * chunk, &acc, 1));
    printf(" the 13 slowest: %.0f GB/s\n", 13 * gb / (time_range(va3, 13 * chunk, sink, e0, e1) * 1e-3));
  }
  // one plain hipMalloc of 13 GiB, per 1-GiB piece and whole
  void* plain;
  if (hipMalloc(&plain, 13 * chunk) == hipSuccess) {
    hipMemset(plain, 0, 13 * chunk);
    hipDeviceSynchronize();
    printf("one hipMalloc of 13 GiB, per 1-GiB piece:\n");
    for (int i = 0; i < 13; ++i) printf("%5.0f ", gb / (time_range((char*)plain + (size_t)i * chunk, chunk, sink, e0, e1) * 1e-3));
    printf("\nwhole: %.0f GB/s\n", 13 * gb / (time_range(plain, 13 * chunk, sink, e0, e1) * 1e-3));
  }
  return 0;
}
