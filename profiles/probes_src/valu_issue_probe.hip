// How many cycles does a wave64 VALU instruction cost a CDNA4 SIMD?  Independent v_fma_f32 / v_cndmask / v_cmp streams at
// 1, 2, 4, 8 wavefronts per SIMD on every CU: prints wave-instructions per SIMD per cycle (clock read from the device
// properties; the run reports the measured time as well).   hipcc --offload-arch=gfx950 -O3 -o valu_issue_probe valu_issue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int KIND>
__global__ __launch_bounds__(256) void k_valu(float* out, int iters, float a, float b)
{
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) {  // 8 independent FMAs
      x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
      x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
    } else if (KIND == 2) {  // 8 plain v_fma_f32 (inline asm: the compiler packs adjacent FMAs into v_pk_fma_f32 otherwise)
#define F(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b))
      F(x0); F(x1); F(x2); F(x3); F(x4); F(x5); F(x6); F(x7);
#undef F
    } else if (KIND == 3) {  // 8 plain v_add_f32
#define F(x) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(a))
      F(x0); F(x1); F(x2); F(x3); F(x4); F(x5); F(x6); F(x7);
#undef F
    } else if (KIND == 4) {  // 4 x (v_cmp_lt_f32 + v_cndmask_b32)
#define F(x, y) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(y) : "vcc")
      F(x0, x1); F(x2, x3); F(x4, x5); F(x6, x7);
#undef F
    } else {          // compare + select pairs (the walk kernels' diet): 4 pairs = 8 instructions
      x0 = x0 < x1 ? x0 + a : x1; x2 = x2 < x3 ? x2 + a : x3; x4 = x4 < x5 ? x4 + a : x5; x6 = x6 < x7 ? x6 + a : x7;
      x1 += b; x3 += b; x5 += b; x7 += b;
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

int main()
{
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  float* out;
  hipMalloc(&out, (size_t)cus * 8 * 256 * sizeof(float) * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 1 << 16;
  printf("%d CUs, clock %.0f MHz\n", cus, prop.clockRate / 1e3);
  const char* names[5] = {"8 FMAs as the compiler packs them (4 v_pk_fma_f32)", "cmp/select/add (12 instructions)", "8 v_fma_f32", "8 v_add_f32", "4 x (v_cmp_lt_f32 + v_cndmask_b32)"};
  const int per_iter[5] = {4, 12, 8, 8, 8};
  for (int kind = 0; kind < 5; ++kind)
    for (int wps : {1, 2, 4, 8}) {
      // blocks of 256 threads = 4 wavefronts = one per SIMD of a CU; wps blocks per CU
      const int blocks = cus * wps;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (kind == 0) hipLaunchKernelGGL(k_valu<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
        else if (kind == 1) hipLaunchKernelGGL(k_valu<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
        else if (kind == 2) hipLaunchKernelGGL(k_valu<2>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
        else if (kind == 3) hipLaunchKernelGGL(k_valu<3>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
        else hipLaunchKernelGGL(k_valu<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
      }
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double instr_per_simd = (double)iters * per_iter[kind] * wps;   // (loop control is scalar)
      printf("%s, %d wavefronts per SIMD: %.3f ms, %.2f cycles per wave-instruction per SIMD at %.0f MHz\n",
             names[kind], wps, ms, ms * 1e-3 * prop.clockRate * 1e3 / instr_per_simd, prop.clockRate / 1e3);
    }
  return 0;
}
