// Developer probe: integrate-like traffic (24 KiB contiguous read + 2 KiB write per block, XCD-chunked
// order) over K read buffers x K write buffers: is the time a property of the read buffer, the write
// buffer or the pair?
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
  const v4* q = lut + (size_t)b * 1536;  // 24 KiB
  v4 acc = {0, 0, 0, 0};
#pragma unroll
  for (int k = 0; k < 12; ++k) acc += __builtin_nontemporal_load(q + k * 128 + threadIdx.x);
  __builtin_nontemporal_store(acc, out + (size_t)b * 128 + threadIdx.x);
}
int main(int argc, char** argv)
{
  const int K = argc > 1 ? atoi(argv[1]) : 3;
  const unsigned ntiles = 64 * 64 * 64;
  const size_t lb = (size_t)ntiles * 24576, ob = (size_t)ntiles * 2048;
  std::vector<v4*> A, B;
  for (int k = 0; k < K; ++k) {
    v4 *a = nullptr, *b = nullptr;
    if (hipMalloc(&b, ob) != hipSuccess || hipMalloc(&a, lb) != hipSuccess) break;
    hipMemset(a, 0, lb);
    A.push_back(a);
    B.push_back(b);
  }
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int round = 0; round < 2; ++round)
    for (size_t i = 0; i < A.size(); ++i) {
      printf("round %d lut %zu:", round, i);
      for (size_t j = 0; j < B.size(); ++j) {
        float t;
        for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k_tile, dim3(ntiles), dim3(128), 0, 0, A[i], B[j], ntiles, 64u);
        hipEventRecord(e0, 0);
        for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(k_tile, dim3(ntiles), dim3(128), 0, 0, A[i], B[j], ntiles, 64u);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&t, e0, e1);
        printf("  %.3f ms (%.0f GB/s)", t / 10, (lb + ob) / (t / 10 * 1e-3) / 1e9);
      }
      printf("\n");
    }
  return 0;
}
