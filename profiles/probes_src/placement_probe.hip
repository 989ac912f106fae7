// Developer probe: does the streaming bandwidth of a buffer depend on where hipMalloc placed it?
// Allocates K buffers of `gb` GB, times a float4 read-sum and a float4 copy on each, several rounds.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_read(const v4* __restrict__ p, size_t n, float* out)
{
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  v4 acc = {0, 0, 0, 0};
  for (; i < n; i += stride) acc += __builtin_nontemporal_load(p + i);
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = 1.0f;
}
__global__ __launch_bounds__(256) void k_read_blocked(const v4* __restrict__ p, size_t n, float* out)
{
  // each block reads one contiguous 64 KiB chunk at a time, blocks advance together (like a tile sweep)
  const size_t per_block = 4096;  // v4 per chunk = 64 KiB
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
  const int K = argc > 1 ? atoi(argv[1]) : 5;
  const double gb = argc > 2 ? atof(argv[2]) : 6.5;
  const size_t bytes = (size_t)(gb * 1e9) / 65536 * 65536, n = bytes / 16;
  std::vector<v4*> bufs;
  float* out;
  hipMalloc(&out, 4);
  for (int k = 0; k < K; ++k) {
    v4* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) { printf("alloc %d failed\n", k); break; }
    hipMemset(p, 0, bytes);
    bufs.push_back(p);
  }
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int round = 0; round < 2; ++round)
    for (size_t k = 0; k < bufs.size(); ++k) {
      float t[2];
      for (int mode = 0; mode < 2; ++mode) {
        for (int w = 0; w < 2; ++w) {
          if (mode == 0) hipLaunchKernelGGL(k_read, dim3(256 * 16), dim3(256), 0, 0, bufs[k], n, out);
          else hipLaunchKernelGGL(k_read_blocked, dim3(256 * 8), dim3(256), 0, 0, bufs[k], n, out);
        }
        hipEventRecord(a, 0);
        for (int r = 0; r < 10; ++r) {
          if (mode == 0) hipLaunchKernelGGL(k_read, dim3(256 * 16), dim3(256), 0, 0, bufs[k], n, out);
          else hipLaunchKernelGGL(k_read_blocked, dim3(256 * 8), dim3(256), 0, 0, bufs[k], n, out);
        }
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        hipEventElapsedTime(&t[mode], a, b);
      }
      printf("round %d buf %zu %p  read %.0f GB/s  blocked %.0f GB/s\n", round, k, (void*)bufs[k], bytes / (t[0] / 10 * 1e-3) / 1e9,
             bytes / (t[1] / 10 * 1e-3) / 1e9);
    }
  return 0;
}
