// sampling.cuh -- device-side texture sampling with the semantics of the
// OpenGL sampler state the reference relies on:
// LINEAR + CLAMP_TO_EDGE (t = s*n - 0.5, blend T0 + a*(T1 - T0), x then y then z)
// and NEAREST + CLAMP_TO_EDGE.  Plain fp32, compiled with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>

namespace rgbdr {

__device__ __forceinline__ float lerpf(float a, float b, float t) { return a + t * (b - a); }

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// floor value -> index; NaN -> -1 (then clamped to texel 0), huge values saturate
__device__ __forceinline__ int idx_from_floor(float f, int n)
{
  return (int)fminf(fmaxf(f, -1.0f), (float)n);
}

struct Axis {
  int i0, i1;
  float a;
};

__device__ __forceinline__ Axis axis_linear(float s, int n)
{
  const float t = s * (float)n - 0.5f;
  const float f = floorf(t);
  const int j = idx_from_floor(f, n);
  Axis r;
  r.a = t - f;
  r.i0 = clampi(j, 0, n - 1);
  r.i1 = clampi(j + 1, 0, n - 1);
  return r;
}

__device__ __forceinline__ int axis_nearest(float s, int n)
{
  return clampi(idx_from_floor(floorf(s * (float)n), n), 0, n - 1);
}

__device__ __forceinline__ float3 lerp3(float3 a, float3 b, float t)
{
  return make_float3(lerpf(a.x, b.x, t), lerpf(a.y, b.y, t), lerpf(a.z, b.z, t));
}
__device__ __forceinline__ float3 xyz_of(float4 v) { return make_float3(v.x, v.y, v.z); }

// trilinear lookup of .xyz in a 16-B-record volume; z index is offset by -zoff
// (slab-resident part of a LUT)
__device__ __forceinline__ float3 tex3d_xyz(const float4* __restrict__ vol, int rx, int ry, int rz, int zoff,
                                            float u, float v, float w)
{
  const Axis X = axis_linear(u, rx), Y = axis_linear(v, ry), Z = axis_linear(w, rz);
  const size_t sy = (size_t)rx, sz = (size_t)rx * ry;
  const float4* z0 = vol + (size_t)(Z.i0 - zoff) * sz;
  const float4* z1 = vol + (size_t)(Z.i1 - zoff) * sz;
  const float3 t000 = xyz_of(z0[Y.i0 * sy + X.i0]), t100 = xyz_of(z0[Y.i0 * sy + X.i1]);
  const float3 t010 = xyz_of(z0[Y.i1 * sy + X.i0]), t110 = xyz_of(z0[Y.i1 * sy + X.i1]);
  const float3 t001 = xyz_of(z1[Y.i0 * sy + X.i0]), t101 = xyz_of(z1[Y.i0 * sy + X.i1]);
  const float3 t011 = xyz_of(z1[Y.i1 * sy + X.i0]), t111 = xyz_of(z1[Y.i1 * sy + X.i1]);
  const float3 c00 = lerp3(t000, t100, X.a), c10 = lerp3(t010, t110, X.a);
  const float3 c01 = lerp3(t001, t101, X.a), c11 = lerp3(t011, t111, X.a);
  return lerp3(lerp3(c00, c10, Y.a), lerp3(c01, c11, Y.a), Z.a);
}

__device__ __forceinline__ float2 tex3d_uv(const float2* __restrict__ vol, int rx, int ry, int rz, float u, float v,
                                           float w)
{
  const Axis X = axis_linear(u, rx), Y = axis_linear(v, ry), Z = axis_linear(w, rz);
  const size_t sy = (size_t)rx, sz = (size_t)rx * ry;
  const float2* z0 = vol + (size_t)Z.i0 * sz;
  const float2* z1 = vol + (size_t)Z.i1 * sz;
  const float2 t000 = z0[Y.i0 * sy + X.i0], t100 = z0[Y.i0 * sy + X.i1];
  const float2 t010 = z0[Y.i1 * sy + X.i0], t110 = z0[Y.i1 * sy + X.i1];
  const float2 t001 = z1[Y.i0 * sy + X.i0], t101 = z1[Y.i0 * sy + X.i1];
  const float2 t011 = z1[Y.i1 * sy + X.i0], t111 = z1[Y.i1 * sy + X.i1];
  float2 r;
  r.x = lerpf(lerpf(lerpf(t000.x, t100.x, X.a), lerpf(t010.x, t110.x, X.a), Y.a),
              lerpf(lerpf(t001.x, t101.x, X.a), lerpf(t011.x, t111.x, X.a), Y.a), Z.a);
  r.y = lerpf(lerpf(lerpf(t000.y, t100.y, X.a), lerpf(t010.y, t110.y, X.a), Y.a),
              lerpf(lerpf(t001.y, t101.y, X.a), lerpf(t011.y, t111.y, X.a), Y.a), Z.a);
  return r;
}

}  // namespace rgbdr
