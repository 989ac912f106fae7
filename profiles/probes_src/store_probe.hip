// Developer probe: cost of the TSDF store stream next to the LUT read stream, by cache policy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v4 __attribute__((ext_vector_type(4)));
template <int MODE>  // 0 nt load + nt store, 1 nt load + plain store, 2 nt load, no store, 3 plain load + nt store, 4: nt load + store every tile but by half the lanes x2 (32 B/lane)
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
  for (int k = 0; k < 12; ++k) acc += MODE == 3 ? q[k * 128 + threadIdx.x] : __builtin_nontemporal_load(q + k * 128 + threadIdx.x);
  if (MODE == 0 || MODE == 3) __builtin_nontemporal_store(acc, out + (size_t)b * 128 + threadIdx.x);
  if (MODE == 1) out[(size_t)b * 128 + threadIdx.x] = acc;
  if (MODE == 2 && acc.x == 1234.5f) out[0] = acc;
}
#define RUN(M) { float t; for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k_tile<M>, dim3(ntiles), dim3(128), 0, 0, A, B, ntiles, 64u); \
  hipEventRecord(e0, 0); for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(k_tile<M>, dim3(ntiles), dim3(128), 0, 0, A, B, ntiles, 64u); \
  hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&t, e0, e1); printf("  mode %d %.3f ms", M, t / 10); }
int main(int argc, char** argv)
{
  const int K = argc > 1 ? atoi(argv[1]) : 4;
  const unsigned ntiles = 64 * 64 * 64;
  const size_t lb = (size_t)ntiles * 24576, ob = (size_t)ntiles * 2048;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int k = 0; k < K; ++k) {
    v4 *A = nullptr, *B = nullptr;
    if (hipMalloc(&B, ob) != hipSuccess || hipMalloc(&A, lb) != hipSuccess) break;
    hipMemset(A, 0, lb);
    printf("buf %d:", k);
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(0)
    printf("\n");
  }
  return 0;
}
