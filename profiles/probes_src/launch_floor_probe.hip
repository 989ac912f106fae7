// What does a dependent launch cost on one stream, and how long is one wavefront's instruction chain?
// (a) chains of 8 launches of a kernel that does nothing but one store, grids of 1 / 64 / 1024 blocks: time per launch from HIP
//     events around the chain (run under rocprofv3 --kernel-trace for the per-kernel durations the profiler reports);
// (b) one launch of `blocks` x 64 threads whose wavefronts run a chain of N dependent v_add_f32 / v_cndmask pairs.
// hipcc --offload-arch=gfx950 -O3 -o launch_floor_probe launch_floor_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void k_empty(float* out) { if (threadIdx.x == 0) out[blockIdx.x] = 1.0f; }

__global__ void k_chain(float* out, int n, float a)
{
  float x = threadIdx.x;
  for (int i = 0; i < n; ++i) {
    asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(a));
    asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(a) : "vcc");
    asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(a));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}

int main()
{
  float* out;
  hipMalloc(&out, 1 << 24);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipStream_t s;
  hipStreamCreate(&s);
  for (int blocks : {1, 64, 1024, 16384}) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0, s);
      for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(k_empty, dim3(blocks), dim3(256), 0, s, out);
      hipEventRecord(e1, s);
      hipStreamSynchronize(s);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep == 2) printf("empty kernel, %5d blocks of 256: %.2f us per launch (8 in a row)\n", blocks, ms * 1e3 / 8);
    }
  }
  // (c) the same chains replayed from a hipGraph (stream capture of 8 / 64 launches): does a graph's dependent launch cost less?
  for (int len : {8, 64})
    for (int blocks : {1, 1024}) {
      hipGraph_t g;
      hipGraphExec_t ge;
      hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
      for (int k = 0; k < len; ++k) hipLaunchKernelGGL(k_empty, dim3(blocks), dim3(256), 0, s, out);
      hipStreamEndCapture(s, &g);
      if (hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) {
        printf("graph instantiate failed\n");
        continue;
      }
      for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0, s);
        for (int k = 0; k < 4; ++k) hipGraphLaunch(ge, s);
        hipEventRecord(e1, s);
        hipStreamSynchronize(s);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep == 3) printf("graph of %2d empty kernels, %5d blocks of 256: %.2f us per kernel (4 graph launches in a row)\n", len, blocks, ms * 1e3 / (4 * len));
      }
      hipGraphExecDestroy(ge);
      hipGraphDestroy(g);
    }
  // (d) ... and behind a kernel that keeps the GPU busy for ~0.1 ms (a graph launch has a fixed cost: does it hide behind the
  // kernel in front of it?): the long kernel alone, + 8 stream launches, + one graph of the same 8
  {
    hipGraph_t g;
    hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(k_empty, dim3(1024), dim3(256), 0, s, out);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int mode = 0; mode < 3; ++mode) {
      float best = 1e9f;
      for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0, s);
        hipLaunchKernelGGL(k_chain, dim3(4096), dim3(64), 0, s, out, 4096, 0.5f);
        if (mode == 1) for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(k_empty, dim3(1024), dim3(256), 0, s, out);
        if (mode == 2) hipGraphLaunch(ge, s);
        hipEventRecord(e1, s);
        hipStreamSynchronize(s);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2 && ms < best) best = ms;
      }
      printf("long kernel %s: %.2f us\n", mode == 0 ? "alone" : mode == 1 ? "+ 8 stream launches (1024 blocks)" : "+ a graph of the 8", best * 1e3);
    }
  }
  for (int blocks : {1, 256, 1024, 4096})
    for (int n : {64, 256, 1024}) {
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, s);
        for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(k_chain, dim3(blocks), dim3(64), 0, s, out, n, 0.5f);
        hipEventRecord(e1, s);
        hipStreamSynchronize(s);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep == 2) printf("chain of %4d instructions, %4d wavefronts: %.2f us per launch (4 in a row)\n", n * 4, blocks, ms * 1e3 / 4);
      }
    }
  return 0;
}
