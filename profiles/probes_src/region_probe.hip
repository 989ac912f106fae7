// Developer probe: read bandwidth of every 128 MiB region of K hipMalloc'ed buffers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_read_blocked(const v4* __restrict__ p, size_t n, float* out)
{
  const size_t per_block = 4096;
  v4 acc = {0, 0, 0, 0};
  for (size_t c = blockIdx.x; c * per_block < n; c += gridDim.x) {
    const v4* q = p + c * per_block;
#pragma unroll 4
    for (int k = threadIdx.x; k < (int)per_block; k += 256) acc += __builtin_nontemporal_load(q + k);
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = 1.0f;
}
int main(int argc, char** argv)
{
  const int K = argc > 1 ? atoi(argv[1]) : 4;
  const double gb = argc > 2 ? atof(argv[2]) : 6.5;
  const size_t region = (size_t)(argc > 3 ? atoi(argv[3]) : 128) << 20;
  const size_t bytes = (size_t)(gb * 1e9) / region * region;
  std::vector<v4*> bufs;
  float* out;
  hipMalloc(&out, 4);
  for (int k = 0; k < K; ++k) {
    v4* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) break;
    hipMemset(p, 0, bytes);
    bufs.push_back(p);
  }
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (size_t k = 0; k < bufs.size(); ++k) {
    float t;
    hipLaunchKernelGGL(k_read_blocked, dim3(2048), dim3(256), 0, 0, bufs[k], bytes / 16, out);
    hipEventRecord(a, 0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k_read_blocked, dim3(2048), dim3(256), 0, 0, bufs[k], bytes / 16, out);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    hipEventElapsedTime(&t, a, b);
    printf("buf %zu %p whole %.0f GB/s | regions:", k, (void*)bufs[k], bytes / (t / 5 * 1e-3) / 1e9);
    for (size_t r0 = 0; r0 < bytes; r0 += region) {
      const v4* q = (const v4*)((const char*)bufs[k] + r0);
      hipLaunchKernelGGL(k_read_blocked, dim3(2048), dim3(256), 0, 0, q, region / 16, out);
      hipEventRecord(a, 0);
      for (int r = 0; r < 8; ++r) hipLaunchKernelGGL(k_read_blocked, dim3(2048), dim3(256), 0, 0, q, region / 16, out);
      hipEventRecord(b, 0);
      hipEventSynchronize(b);
      hipEventElapsedTime(&t, a, b);
      printf(" %.1f", region / (t / 8 * 1e-3) / 1e12);
    }
    printf("\n");
  }
  return 0;
}
