// kernels_raymarch.hip -- TSDF ray-marcher, the consumer of the
// volume: glsl/tsdf_raymarch.{vs,fs} + glsl/shading.glsl driven by
// ReconIntegration::draw (framework/reconstruction/recon_integration.cpp:177-241).
//
// One thread per viewport pixel (16x16 pixel blocks: neighbouring rays touch
// neighbouring tiles).  The reference rasterises the unit cube and every fragment
// of a pixel marches the same ray from CameraPos; here the ray is built from the
// pixel centre (screenToVol of the far-plane point, tsdf_raymarch.fs:384-390) and a
// pixel is covered when its ray meets the unit cube in front of the camera.  The
// volume is sampled LINEAR + CLAMP_TO_EDGE straight from the tile-linear layout
// (sampler unit 29 in the reference), the per-sensor colour blend samples the
// resident inverse LUT (grid layout, or the file volume with
// RGBDR_FLAG_NO_RESAMPLE), cv_uv, the RGB8 colour frames and the packed
// depth/quality frames.  Latency / gather bound; reported as time.
#include <cstdlib>

#include <hip/hip_runtime.h>

#include "dxt.cuh"
#include "rgbdr_internal.hpp"
#include "sampling.cuh"

namespace rgbdr {

__device__ __forceinline__ float4 mat4_mul(const float* m, float x, float y, float z, float w)
{
  float4 o;
  o.x = (m[0] * x + m[4] * y) + (m[8] * z + m[12] * w);
  o.y = (m[1] * x + m[5] * y) + (m[9] * z + m[13] * w);
  o.z = (m[2] * x + m[6] * y) + (m[10] * z + m[14] * w);
  o.w = (m[3] * x + m[7] * y) + (m[11] * z + m[15] * w);
  return o;
}

// tile-linear fetch; `tz_alloc0` = global index of the first resident tile layer (a Z slab
// keeps halo layers below / above the layers it owns; 0 for a whole volume)
__device__ __forceinline__ float tile_at(const float* __restrict__ v, int TX, int TY, int tz_alloc0, int x, int y, int z)
{
  return v[((size_t)(((z >> 3) - tz_alloc0) * TY + (y >> 3)) * TX + (x >> 3)) * kTileVoxels + ((z & 7) * 64 + (y & 7) * 8 + (x & 7))];
}

// A Z slab holds voxel rows [res_z0, res_z1) only.  Every sample the slab protocol needs lies
// inside them.  A refined position that is NaN (the TSDF holds NaN where the
// reference's weighting divides 0 by 0) selects texel 0 in GL; its weights are NaN too, so the
// value read does not matter and the nearest resident row is read instead -- as it is with
// stale halos -- rather than faulting.  No-op for a whole volume.
__device__ __forceinline__ Axis resident_rows(const RaymarchParams& p, Axis z)
{
  z.i0 = clampi(z.i0, p.res_z0, p.res_z1 - 1);
  z.i1 = clampi(z.i1, p.res_z0, p.res_z1 - 1);
  return z;
}

// the eight texels of texture(volume_tsdf, pos) and their blend, apart: march_ahead issues the loads of all its samples
// before it blends any (a blend inside the branch that skips a known sample would wait for that sample's loads alone)
struct Taps {
  float t[8];
};
__device__ __forceinline__ Taps tsdf_taps(const RaymarchParams& p, const Axis& X, const Axis& Y, const Axis& Z)
{
  // tile-linear address = a sum of one term per axis: six terms for the eight texels
  const unsigned x0 = (unsigned)((X.i0 >> 3) * kTileVoxels + (X.i0 & 7)), x1 = (unsigned)((X.i1 >> 3) * kTileVoxels + (X.i1 & 7));
  const unsigned sy = (unsigned)p.TX * kTileVoxels;
  const unsigned y0 = (unsigned)(Y.i0 >> 3) * sy + (unsigned)((Y.i0 & 7) * 8), y1 = (unsigned)(Y.i1 >> 3) * sy + (unsigned)((Y.i1 & 7) * 8);
  const size_t sz = (size_t)p.TY * sy;
  const float* z0 = p.tsdf + (size_t)((Z.i0 >> 3) - p.tz_alloc0) * sz + (size_t)((Z.i0 & 7) * 64);
  const float* z1 = p.tsdf + (size_t)((Z.i1 >> 3) - p.tz_alloc0) * sz + (size_t)((Z.i1 & 7) * 64);
  Taps r;
  r.t[0] = z0[y0 + x0], r.t[1] = z0[y0 + x1];
  r.t[2] = z0[y1 + x0], r.t[3] = z0[y1 + x1];
  r.t[4] = z1[y0 + x0], r.t[5] = z1[y0 + x1];
  r.t[6] = z1[y1 + x0], r.t[7] = z1[y1 + x1];
  return r;
}
__device__ __forceinline__ float tsdf_blend(const Taps& r, float ax, float ay, float az)
{
  return lerpf(lerpf(lerpf(r.t[0], r.t[1], ax), lerpf(r.t[2], r.t[3], ax), ay), lerpf(lerpf(r.t[4], r.t[5], ax), lerpf(r.t[6], r.t[7], ax), ay), az);
}

// texture(volume_tsdf, pos).r
__device__ __forceinline__ float tsdf_sample(const RaymarchParams& p, float px, float py, float pz)
{
  const Axis X = axis_linear(px, p.X), Y = axis_linear(py, p.Y);
  const Axis Z = resident_rows(p, axis_linear(pz, p.Z));
  const int a0 = p.tz_alloc0;
  const float t000 = tile_at(p.tsdf, p.TX, p.TY, a0, X.i0, Y.i0, Z.i0), t100 = tile_at(p.tsdf, p.TX, p.TY, a0, X.i1, Y.i0, Z.i0);
  const float t010 = tile_at(p.tsdf, p.TX, p.TY, a0, X.i0, Y.i1, Z.i0), t110 = tile_at(p.tsdf, p.TX, p.TY, a0, X.i1, Y.i1, Z.i0);
  const float t001 = tile_at(p.tsdf, p.TX, p.TY, a0, X.i0, Y.i0, Z.i1), t101 = tile_at(p.tsdf, p.TX, p.TY, a0, X.i1, Y.i0, Z.i1);
  const float t011 = tile_at(p.tsdf, p.TX, p.TY, a0, X.i0, Y.i1, Z.i1), t111 = tile_at(p.tsdf, p.TX, p.TY, a0, X.i1, Y.i1, Z.i1);
  return lerpf(lerpf(lerpf(t000, t100, X.a), lerpf(t010, t110, X.a), Y.a),
               lerpf(lerpf(t001, t101, X.a), lerpf(t011, t111, X.a), Y.a), Z.a);
}

// the voxel row the LINEAR footprint of a sample starts at decides which Z slab owns it
__device__ __forceinline__ bool sample_owned(const RaymarchParams& p, float pz)
{
  const int z0 = axis_linear(pz, p.Z).i0;
  return z0 >= p.own_z0 && z0 < p.own_z1;
}

// texture(cv_xyz_inv[i], pos).xyz from the grid-layout planes ([tile][N][3][512])
__device__ __forceinline__ float3 lut_planes_sample(const RaymarchParams& p, int sensor, float px, float py, float pz)
{
  const Axis X = axis_linear(px, p.X), Y = axis_linear(py, p.Y);
  const Axis Z = resident_rows(p, axis_linear(pz, p.Z));
  float r[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    auto at = [&](int x, int y, int z) {
      const size_t tile = (size_t)(((z >> 3) - p.tz_alloc0) * p.TY + (y >> 3)) * p.TX + (x >> 3);
      return p.lut_tiled[((tile * p.N + sensor) * 3 + c) * kTileVoxels + ((z & 7) * 64 + (y & 7) * 8 + (x & 7))];
    };
    r[c] = lerpf(lerpf(lerpf(at(X.i0, Y.i0, Z.i0), at(X.i1, Y.i0, Z.i0), X.a), lerpf(at(X.i0, Y.i1, Z.i0), at(X.i1, Y.i1, Z.i0), X.a), Y.a),
                 lerpf(lerpf(at(X.i0, Y.i0, Z.i1), at(X.i1, Y.i0, Z.i1), X.a), lerpf(at(X.i0, Y.i1, Z.i1), at(X.i1, Y.i1, Z.i1), X.a), Y.a),
                 Z.a);
  }
  return make_float3(r[0], r[1], r[2]);
}

__device__ __forceinline__ float3 normalize3(float x, float y, float z)
{
  const float l = sqrtf(x * x + y * y + z * z);
  return make_float3(x / l, y / l, z / l);
}

__constant__ float c_camera_colors[5][3] = {{228, 26, 28}, {55, 126, 184}, {77, 175, 74}, {152, 78, 163}, {255, 127, 0}};

struct Ray {
  float sp[3], step[3];
  unsigned max_num;
  bool covered;
};

// ray of a pixel: direction, first sample position and sample budget (main(), :62-86)
__device__ __forceinline__ Ray ray_setup(const RaymarchParams& p, int px, int py, size_t o)
{
  Ray r;
  r.covered = false;
  r.max_num = 0u;
  const float sd = p.limit * 0.5f;
  const float4 pc = mat4_mul(p.img_to_eye, (float)px + 0.5f, (float)py + 0.5f, 1.0f, 1.0f);
  const float4 ws = mat4_mul(p.modelview_inv, pc.x / pc.w, pc.y / pc.w, pc.z / pc.w, 1.0f);
  const float4 tv = mat4_mul(p.vol_to_world_inv, ws.x, ws.y, ws.z, ws.w);
  const float3 nd = normalize3(tv.x - p.camera_pos[0], tv.y - p.camera_pos[1], tv.z - p.camera_pos[2]);
  r.step[0] = nd.x * sd;
  r.step[1] = nd.y * sd;
  r.step[2] = nd.z * sd;
  float tmin[3], tmax[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float inv = 1.0f / r.step[a];
    const float tb = inv * (0.0f - p.camera_pos[a]), tt = inv * (1.0f - p.camera_pos[a]);
    tmin[a] = fminf(tt, tb);
    tmax[a] = fmaxf(tt, tb);
  }
  const float t0 = fmaxf(fmaxf(tmin[0], tmin[1]), fmaxf(tmin[0], tmin[2]));
  const float t1 = fminf(fminf(tmax[0], tmax[1]), fminf(tmax[0], tmax[2]));
  if (!(t0 <= t1) || !(t1 > 0.0f)) return r;  // cube not rasterised onto this pixel
  r.covered = true;
  const float t_near = t0 < 0.0f ? 0.0f : t0;
#pragma unroll
  for (int a = 0; a < 3; ++a) r.sp[a] = p.camera_pos[a] + r.step[a] * t_near;
  float fmaxs = ceilf(fabsf(t1 - t_near));
  if (p.skip_space) {  // getStartPos, tsdf_raymarch.fs:392-401
    const float4 dmm = p.peels[o];
    const float dr = (dmm.x >= dmm.z) ? 0.0f : dmm.x;  // closest back face is the closest face -> gl_DepthRange.near
    float pf[3], pb[3];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const float4 a4 = mat4_mul(p.img_to_eye, (float)px + 0.5f, (float)py + 0.5f, k == 0 ? dr : -dmm.y, 1.0f);
      const float4 w4 = mat4_mul(p.modelview_inv, a4.x / a4.w, a4.y / a4.w, a4.z / a4.w, 1.0f);
      const float4 v4 = mat4_mul(p.vol_to_world_inv, w4.x, w4.y, w4.z, w4.w);
      float* dst = k == 0 ? pf : pb;
      dst[0] = v4.x;
      dst[1] = v4.y;
      dst[2] = v4.z;
    }
    if (dr >= 1.0f) {
      pb[0] = pf[0];
      pb[1] = pf[1];
      pb[2] = pf[2];
    }
    r.sp[0] = pf[0];
    r.sp[1] = pf[1];
    r.sp[2] = pf[2];
    const float ex = pf[0] - pb[0], ey = pf[1] - pb[1], ez = pf[2] - pb[2];
    fmaxs = ceilf(sqrtf(ex * ex + ey * ey + ez * ez) / sd);
  }
  r.max_num = !(fmaxs > 0.0f) ? 0u : (fmaxs >= 2147483520.0f ? 2147483520u : (unsigned)fmaxs);
  return r;
}

// submitFragment (:116-142) at the refined position + the depth clamp / GL_LESS test
__device__ __forceinline__ void shade_fragment(const RaymarchParams& p, const float* sp, float4& rgba, float& fdepth)
{
  const float limit = p.limit, sd = limit * 0.5f;
  const float gx = tsdf_sample(p, sp[0] + sd, sp[1], sp[2]) - tsdf_sample(p, sp[0] - sd, sp[1], sp[2]);
  const float gy = tsdf_sample(p, sp[0], sp[1] + sd, sp[2]) - tsdf_sample(p, sp[0], sp[1] - sd, sp[2]);
  const float gz = tsdf_sample(p, sp[0], sp[1], sp[2] + sd) - tsdf_sample(p, sp[0], sp[1], sp[2] - sd);
  const float3 gn = normalize3(gx, gy, gz);
  const float4 vn4 = mat4_mul(p.normal_matrix, -gn.x, -gn.y, -gn.z, 0.0f);
  const float3 vn = normalize3(vn4.x, vn4.y, vn4.z);
  const float4 vp = mat4_mul(p.mv_vol_to_world, sp[0], sp[1], sp[2], 1.0f);
  float tc[3] = {0, 0, 0}, tc2[3] = {0, 0, 0}, tw = 0.0f, tw2 = 0.0f, cw[3] = {0, 0, 0}, cwt = 0.0f;
  // blendColors' loop over the sensors, four at a time, as three rounds of gathers.  A sensor's lookups depend on each other
  // (calibration volume -> colour lookup volume -> colour texels; calibration volume -> depth -> quality), the sensors' do not,
  // and a wavefront that shades spends its time waiting: ~17 dependent round trips to memory at two wavefronts per SIMD
  // (profiles/r06_notes/raymarch.md).  So every gather of a round -- for all four sensors -- is issued before any of their
  // results is looked at (the scheduling barriers keep the compiler from sinking the loads back to their uses): three round
  // trips per group of four sensors.  Nothing is loaded behind a branch (a sensor index past N reads sensor N - 1 again,
  // the quality texels are read whatever the depth test says; every address is clamped into its image); the sums are
  // formed in the shader's order.
  constexpr int G = 4;
  const Axis LX = axis_linear(sp[0], p.X), LY = axis_linear(sp[1], p.Y), LZ = resident_rows(p, axis_linear(sp[2], p.Z));
  for (int i0 = 0; i0 < p.N; i0 += G) {
    int sensor[G];
#pragma unroll
    for (int g = 0; g < G; ++g) sensor[g] = min(i0 + g, p.N - 1);
    // round 1: texture(cv_xyz_inv[i], pos).xyz -- eight texels of three planes (or eight 16-byte records) per sensor
    float lt[G][3][8];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (p.lut_tiled) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int x = (k & 1) ? LX.i1 : LX.i0, y = (k & 2) ? LY.i1 : LY.i0, z = (k & 4) ? LZ.i1 : LZ.i0;
          const size_t tile = (size_t)(((z >> 3) - p.tz_alloc0) * p.TY + (y >> 3)) * p.TX + (x >> 3);
          const float* at = p.lut_tiled + ((tile * p.N + sensor[g]) * 3) * kTileVoxels + ((z & 7) * 64 + (y & 7) * 8 + (x & 7));
#pragma unroll
          for (int c = 0; c < 3; ++c) lt[g][c][k] = at[c * kTileVoxels];
        }
      } else {
        const int i = sensor[g];
        const Axis X = axis_linear(sp[0], p.rx[i]), Y = axis_linear(sp[1], p.ry[i]), Z = axis_linear(sp[2], p.rz[i]);
        const size_t sy = (size_t)p.rx[i], sz = (size_t)p.rx[i] * p.ry[i];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float4 v = p.lut[i][(size_t)(((k & 4) ? Z.i1 : Z.i0) - p.zoff[i]) * sz + ((k & 2) ? Y.i1 : Y.i0) * sy + ((k & 1) ? X.i1 : X.i0)];
          lt[g][0][k] = v.x, lt[g][1][k] = v.y, lt[g][2][k] = v.z;
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    float3 pcal[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      float r[3];
      if (p.lut_tiled) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
          r[c] = lerpf(lerpf(lerpf(lt[g][c][0], lt[g][c][1], LX.a), lerpf(lt[g][c][2], lt[g][c][3], LX.a), LY.a),
                       lerpf(lerpf(lt[g][c][4], lt[g][c][5], LX.a), lerpf(lt[g][c][6], lt[g][c][7], LX.a), LY.a), LZ.a);
      } else {  // (tex3d_xyz's order: the x blends of the three components, then y, then z)
        const int i = sensor[g];
        const Axis X = axis_linear(sp[0], p.rx[i]), Y = axis_linear(sp[1], p.ry[i]), Z = axis_linear(sp[2], p.rz[i]);
#pragma unroll
        for (int c = 0; c < 3; ++c)
          r[c] = lerpf(lerpf(lerpf(lt[g][c][0], lt[g][c][1], X.a), lerpf(lt[g][c][2], lt[g][c][3], X.a), Y.a),
                       lerpf(lerpf(lt[g][c][4], lt[g][c][5], X.a), lerpf(lt[g][c][6], lt[g][c][7], X.a), Y.a), Z.a);
      }
      pcal[g] = make_float3(r[0], r[1], r[2]);
    }
    // round 2: texture(cv_uv[i], pcal) -- eight float2 texels -- and the depth and quality texels of the sensor's frame
    float2 ut[G][8];
    Axis UX[G], UY[G], UZ[G], QX[G], QY[G];
    float depth[G], q4[G][4];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int i = sensor[g];
      UX[g] = axis_linear(pcal[g].x, p.uv_res[i][0]), UY[g] = axis_linear(pcal[g].y, p.uv_res[i][1]), UZ[g] = axis_linear(pcal[g].z, p.uv_res[i][2]);
      const size_t sy = (size_t)p.uv_res[i][0], sz = (size_t)p.uv_res[i][0] * p.uv_res[i][1];
#pragma unroll
      for (int k = 0; k < 8; ++k)
        ut[g][k] = p.cv_uv[i][(size_t)((k & 4) ? UZ[g].i1 : UZ[g].i0) * sz + ((k & 2) ? UY[g].i1 : UY[g].i0) * sy + ((k & 1) ? UX[g].i1 : UX[g].i0)];
      const uint2* frame = p.frame[i];
      const int ix = axis_nearest(pcal[g].x, p.W), iy = axis_nearest(pcal[g].y, p.H);
      depth[g] = __uint_as_float(frame[(size_t)iy * p.W + ix].x);
      QX[g] = axis_linear(pcal[g].x, p.W), QY[g] = axis_linear(pcal[g].y, p.H);
      q4[g][0] = __uint_as_float(frame[(size_t)QY[g].i0 * p.W + QX[g].i0].y & 0x7fffffffu);
      q4[g][1] = __uint_as_float(frame[(size_t)QY[g].i0 * p.W + QX[g].i1].y & 0x7fffffffu);
      q4[g][2] = __uint_as_float(frame[(size_t)QY[g].i1 * p.W + QX[g].i0].y & 0x7fffffffu);
      q4[g][3] = __uint_as_float(frame[(size_t)QY[g].i1 * p.W + QX[g].i1].y & 0x7fffffffu);
    }
    __builtin_amdgcn_sched_barrier(0);
    float2 pcol[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const float ax = UX[g].a, ay = UY[g].a, az = UZ[g].a;
      pcol[g].x = lerpf(lerpf(lerpf(ut[g][0].x, ut[g][1].x, ax), lerpf(ut[g][2].x, ut[g][3].x, ax), ay),
                        lerpf(lerpf(ut[g][4].x, ut[g][5].x, ax), lerpf(ut[g][6].x, ut[g][7].x, ax), ay), az);
      pcol[g].y = lerpf(lerpf(lerpf(ut[g][0].y, ut[g][1].y, ax), lerpf(ut[g][2].y, ut[g][3].y, ax), ay),
                        lerpf(lerpf(ut[g][4].y, ut[g][5].y, ax), lerpf(ut[g][6].y, ut[g][7].y, ax), ay), az);
    }
    // round 3: the four colour texels around pcol, from the frame's DXT blocks as uploaded or from the decoded frame
    float col[G][3];
    Axis CX[G], CY[G];
#pragma unroll
    for (int g = 0; g < G; ++g) CX[g] = axis_linear(pcol[g].x, p.Wc), CY[g] = axis_linear(pcol[g].y, p.Hc);
    if (p.color_dxt) {
      uint2 blk[G][4];
      const int bw = (p.Wc + 3) / 4;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const uint8_t* layer = p.color_dxt + (size_t)sensor[g] * p.color_layer_bytes;
#pragma unroll
        for (int k = 0; k < 4; ++k) blk[g][k] = dxt_block(layer, bw, p.color_mode, ((k & 1) ? CX[g].i1 : CX[g].i0) >> 2, ((k & 2) ? CY[g].i1 : CY[g].i0) >> 2);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < G; ++g) {
        int t4[4][3];
#pragma unroll
        for (int k = 0; k < 4; ++k) dxt_texel(blk[g][k], p.color_mode, (k & 1) ? CX[g].i1 : CX[g].i0, (k & 2) ? CY[g].i1 : CY[g].i0, t4[k]);
#pragma unroll
        for (int k = 0; k < 3; ++k)
          col[g][k] = lerpf(lerpf((float)t4[0][k] / 255.0f, (float)t4[1][k] / 255.0f, CX[g].a), lerpf((float)t4[2][k] / 255.0f, (float)t4[3][k] / 255.0f, CX[g].a),
                            CY[g].a);
      }
    } else {
      uint8_t c4[G][4][3];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const uint8_t* img = p.color + (size_t)sensor[g] * p.Wc * p.Hc * 3;
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int c = 0; c < 3; ++c) c4[g][k][c] = img[((size_t)((k & 2) ? CY[g].i1 : CY[g].i0) * p.Wc + ((k & 1) ? CX[g].i1 : CX[g].i0)) * 3 + c];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int k = 0; k < 3; ++k)
          col[g][k] = lerpf(lerpf((float)c4[g][0][k] / 255.0f, (float)c4[g][1][k] / 255.0f, CX[g].a),
                            lerpf((float)c4[g][2][k] / 255.0f, (float)c4[g][3][k] / 255.0f, CX[g].a), CY[g].a);
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int i = i0 + g;
      if (i >= p.N) break;
      const float dist = fabsf(depth[g] - pcal[g].z);
      const float q = dist < limit ? lerpf(lerpf(q4[g][0], q4[g][1], QX[g].a), lerpf(q4[g][2], q4[g][3], QX[g].a), QY[g].a) : 0.0f;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        tc[k] += col[g][k] * q / (dist + 0.01f);
        tc2[k] += col[g][k] / dist;
        if (i < 5) cw[k] += (c_camera_colors[i][k] / 255.0f) * q;
      }
      tw += q / (dist + 0.01f);
      tw2 += 1.0f / dist;
      cwt += q;
    }
  }
  if (p.shade_mode == 3) {
    rgba = make_float4(cwt <= 0.0f ? 1.0f : cw[0] / cwt, cwt <= 0.0f ? 1.0f : cw[1] / cwt, cwt <= 0.0f ? 1.0f : cw[2] / cwt, 1.0f);
  } else {
    float diff[4];
    if (tw > 0.0f) {
      diff[0] = tc[0] / tw;
      diff[1] = tc[1] / tw;
      diff[2] = tc[2] / tw;
      diff[3] = 1.0f;
    } else {
      diff[0] = tc2[0] / tw2;
      diff[1] = tc2[1] / tw2;
      diff[2] = tc2[2] / tw2;
      diff[3] = -1.0f;
    }
    float r[3] = {1.0f, 1.0f, 1.0f};
    if (p.shade_mode == 0) {
      r[0] = diff[0];
      r[1] = diff[1];
      r[2] = diff[2];
    } else if (p.shade_mode == 1) {
      const float3 tln = normalize3(1.5f - vp.x, 1.0f - vp.y, 1.0f - vp.z);
      const float la = vn.x * tln.x + vn.y * tln.y + vn.z * tln.z;
      float dc = 0.0f, sl = 0.0f;
      if (!(la <= 0.0f)) {
        dc = fmaxf(la, 0.0f);
        const float3 tvw = normalize3(-vp.x, -vp.y, -vp.z);
        const float3 hn = normalize3(tln.x + tvw.x, tln.y + tvw.y, tln.z + tvw.z);
        const float ra = hn.x * vn.x + hn.y * vn.y + hn.z * vn.z;
        const float r2 = ra * ra, r4 = r2 * r2, r8 = r4 * r4, r16 = r8 * r8;
        sl = r16 * r4;
        const float a = (1.0f - la) * (1.0f - la);
        sl *= 1.0f - a * a * a;
      }
      const float ld[3] = {1.0f, 0.9f, 0.7f};
#pragma unroll
      for (int k = 0; k < 3; ++k) r[k] = (ld[k] * 0.2f) * 0.5f + ld[k] * 0.5f * dc + 1.0f * 0.5f * sl;
    } else if (p.shade_mode == 2) {
      const float4 r4 = mat4_mul(p.gl_normal_matrix_inv, vn.x, vn.y, vn.z, 0.0f);
      r[0] = r4.x;
      r[1] = r4.y;
      r[2] = r4.z;
    }
    rgba = make_float4(r[0], r[1], r[2], diff[3]);
  }
  // gl_FragDepth is clamped to the depth range and tested GL_LESS against the cleared 1.0
  const float fd = (p.projection[10] * vp.z + p.projection[14]) / -vp.z * 0.5f + 0.5f;
  fdepth = fminf(fmaxf(fd, 0.0f), 1.0f);
  if (!(fdepth < 1.0f)) {
    rgba = make_float4(0.0f, 1.0f, 0.0f, 0.0f);
    fdepth = 1.0f;
  }
}

// The march of main() (tsdf_raymarch.fs:88-110).  The sample positions do not depend on what is sampled, so kAhead
// samples are fetched together and then looked at in the shader's order (the loop's only dependence on a load is the
// break): the positions are accumulated by the same additions, a sample fetched behind the hit or behind the budget is
// simply not looked at (LINEAR + CLAMP_TO_EDGE addresses are always inside the volume).  Rays that start at the brick
// peels are a few samples long and wait for each fetch: eight at a time (4: -7 %, 8: -35 %, 12 spills); rays through the whole cube are bound by the
// number of fetches, not by their latency: one at a time.
#ifndef RGBDR_AHEAD_SKIP
#define RGBDR_AHEAD_SKIP 8
#endif
#ifndef RGBDR_AHEAD_FULL
#define RGBDR_AHEAD_FULL 1
#endif
// Tiles known to hold -limit throughout (p.empty_tiles, k_empty_tiles): a sample whose 2 x 2 x 2 footprint lies in such tiles
// is T0 + a * (T1 - T0) of eight equal finite values -- exactly -limit for every weight -- so it is not fetched.  Its
// position is still accumulated and it is still counted: the same additions, the same sample count, the same frame.
template <int kAhead>
__device__ __forceinline__ void march_ahead(const RaymarchParams& p, const Ray& r, float* sp, float& prev, unsigned& num, bool& hit,
                                            const unsigned* empty_bits = nullptr, int max_rounds = 0x7fffffff)
{
  const float held = -p.limit;
  for (int round = 0; round < max_rounds && num < r.max_num && !hit; ++round) {
    float pos[kAhead][3], dens[kAhead];
    if (empty_bits) {
      Axis X[kAhead], Y[kAhead], Z[kAhead];
      bool known[kAhead];
#pragma unroll
      for (int j = 0; j < kAhead; ++j) {
#pragma unroll
        for (int a = 0; a < 3; ++a) pos[j][a] = j == 0 ? sp[a] : pos[j - 1][a] + r.step[a];
        X[j] = axis_linear(pos[j][0], p.X);
        Y[j] = axis_linear(pos[j][1], p.Y);
        Z[j] = resident_rows(p, axis_linear(pos[j][2], p.Z));
        const unsigned tile = (unsigned)(((Z[j].i0 >> 3) * p.TY + (Y[j].i0 >> 3)) * p.TX + (X[j].i0 >> 3));
        known[j] = ((empty_bits[tile >> 5] >> (tile & 31u)) & 1u) != 0u;
      }
      Taps taps[kAhead];
#pragma unroll
      for (int j = 0; j < kAhead; ++j) {
        if (!known[j]) {
          taps[j] = tsdf_taps(p, X[j], Y[j], Z[j]);
        } else {
#pragma unroll
          for (int k = 0; k < 8; ++k) taps[j].t[k] = held;  // (the blend of eight equal values is that value)
        }
      }
#pragma unroll
      for (int j = 0; j < kAhead; ++j) dens[j] = tsdf_blend(taps[j], X[j].a, Y[j].a, Z[j].a);
    } else {
#pragma unroll
      for (int j = 0; j < kAhead; ++j) {
#pragma unroll
        for (int a = 0; a < 3; ++a) pos[j][a] = j == 0 ? sp[a] : pos[j - 1][a] + r.step[a];
        dens[j] = tsdf_sample(p, pos[j][0], pos[j][1], pos[j][2]);
      }
    }
#pragma unroll
    for (int j = 0; j < kAhead; ++j) {
      if (hit || !(num < r.max_num)) break;
      num += 1u;
      const float density = dens[j];
      if (density > 0.0f) {
        const float f = prev / (density - prev);
#pragma unroll
        for (int a = 0; a < 3; ++a) sp[a] = (pos[j][a] - r.step[a]) - r.step[a] * f;
        hit = true;
      } else {
        prev = density;
#pragma unroll
        for (int a = 0; a < 3; ++a) sp[a] = pos[j][a] + r.step[a];
      }
    }
  }
}

// p.empty_bits: a bit per tile = the tile and its +1 neighbours along x, y and z (a LINEAR footprint that starts in the tile
// ends in one of those) have held -limit since a sweep of the current epoch (IntegrateParams::tile_state, k_brick_clear).
// 32 KiB for a 512^3 volume: every workgroup of the march that has a ray to march keeps a copy in LDS.
__device__ __forceinline__ void empty_tiles_block(const unsigned* __restrict__ tile_state, unsigned epoch, int TX, int TY, int TZ,
                                                  unsigned long long* __restrict__ bits, unsigned block, unsigned lane256)
{
  const unsigned t = block * 256u + lane256;
  bool all = false;
  if (t < (unsigned)(TX * TY * TZ)) {
    const int tx = t % TX, ty = (t / TX) % TY, tz = t / (TX * TY);
    const int x1 = min(tx + 1, TX - 1), y1 = min(ty + 1, TY - 1), z1 = min(tz + 1, TZ - 1);
    all = true;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int x = (k & 1) ? x1 : tx, y = (k & 2) ? y1 : ty, z = (k & 4) ? z1 : tz;
      all = all && tile_state[((size_t)z * TY + y) * TX + x] == epoch;
    }
  }
  const unsigned long long m = __ballot(all);   // (tiles past the end: 0)
  if ((lane256 & 63u) == 0) bits[t >> 6] = m;
}
__global__ __launch_bounds__(256) void k_empty_tiles(const unsigned* __restrict__ tile_state, unsigned epoch, int TX, int TY, int TZ,
                                                     unsigned long long* __restrict__ bits)
{
  empty_tiles_block(tile_state, epoch, TX, TY, TZ, bits, blockIdx.x, threadIdx.x);
}
void launch_empty_tiles(const unsigned* tile_state, unsigned epoch, int TX, int TY, int TZ, unsigned* bits, hipStream_t s)
{
  const unsigned n = (unsigned)(TX * TY * TZ);
  hipLaunchKernelGGL(k_empty_tiles, dim3((n + 255) / 256), dim3(256), 0, s, tile_state, epoch, TX, TY, TZ, (unsigned long long*)bits);
}

constexpr int kNoHit = 0x7fffffff;

// MODE 0: whole volume in this context, march + shade.
// MODE 1 (Z slab, "find"): index of the first sample this slab owns whose density is > 0.
// MODE 2 (Z slab, "shade"): with the minimum of those indices over all slabs in khit, the
//         slab that owns that sample refines and shades it; the others mark the pixel kNoHit.
// The whole wavefront marches the ray of ONE of its lanes, `owner`: lane j takes sample j of the ray's next 64.  The
// positions are the march's own chain of additions (lane j runs j of them), a sample past the ray's budget is not fetched,
// the first lane with a positive density is the hit and the lane before it (or the round before) holds `prev`: the same
// values in the same order as march_ahead, 64 samples per round trip to memory instead of kAhead.  All 64 lanes call it.
__device__ __forceinline__ void march_whole_wave(const RaymarchParams& p, const Ray& r, int owner, int lane, float* sp, float& prev, unsigned& num,
                                                 bool& hit)
{
  auto from_owner = [&](float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), owner)); };
  float osp[3] = {from_owner(sp[0]), from_owner(sp[1]), from_owner(sp[2])};
  const float ostep[3] = {from_owner(r.step[0]), from_owner(r.step[1]), from_owner(r.step[2])};
  float oprev = from_owner(prev);
  unsigned onum = (unsigned)__builtin_amdgcn_readlane((int)num, owner);
  const unsigned omax = (unsigned)__builtin_amdgcn_readlane((int)r.max_num, owner);
  bool ohit = false;
  while (onum < omax && !ohit) {
    float pos[3] = {osp[0], osp[1], osp[2]};
    for (int k = 0; k < 63; ++k) {
      const bool on = k < lane;
#pragma unroll
      for (int a = 0; a < 3; ++a) pos[a] = on ? pos[a] + ostep[a] : pos[a];
    }
    const unsigned left = omax - onum;
    const bool valid = (unsigned)lane < left;
    float dens = 0.0f;
    if (valid) dens = tsdf_sample(p, pos[0], pos[1], pos[2]);
    const unsigned long long hits = __ballot(valid && dens > 0.0f);
    auto from_lane = [&](float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); };
    if (hits) {
      const int f = __ffsll((long long)hits) - 1;
      const float density = from_lane(dens, f), before = f == 0 ? oprev : from_lane(dens, f - 1);
      const float fr = before / (density - before);
#pragma unroll
      for (int a = 0; a < 3; ++a) osp[a] = (from_lane(pos[a], f) - ostep[a]) - ostep[a] * fr;
      onum += (unsigned)f + 1u;
      ohit = true;
    } else {
      const int last = (int)(left < 64u ? left : 64u) - 1;
      oprev = from_lane(dens, last);
#pragma unroll
      for (int a = 0; a < 3; ++a) osp[a] = from_lane(pos[a], last) + ostep[a];
      onum += (unsigned)last + 1u;
    }
  }
  if (lane == owner) {
    sp[0] = osp[0], sp[1] = osp[1], sp[2] = osp[2];
    prev = oprev;
    num = onum;
    hit = ohit;
  }
}

// A wavefront covers an 8 x 8 pixel square of the block's 16 x 16 (not 16 x 4 rows of the launch order): its rays stay closer
// together in the volume, so a gather instruction touches fewer cache lines -- the march is bound by the vector L1's rate
// of one line per cycle per CU (profiles/r04_notes: 457 M line accesses per 1280 x 720 frame = 0.74 ms).
__device__ __forceinline__ void wave_square_pixel(int& px, int& py)
{
  const int t = threadIdx.y * 16 + threadIdx.x, w = t >> 6, l = t & 63;
  px = blockIdx.x * 16 + (w & 1) * 8 + (l & 7);   // (Morton order of the lanes inside the square: no further gain)
  py = blockIdx.y * 16 + (w >> 1) * 8 + (l >> 3);
}

#ifdef RGBDR_TRACE_BLOCKS
// Developer build (make trace, profiles/march_waves_probe.py): when every wavefront of the whole-volume march went through its
// stages.  Per wavefront {s_memrealtime (100 MHz) at entry, after the ray set-up, after the march, at exit, rounds marched lane
// by lane, rays taken by the whole wavefront, HW_ID, XCC_ID}.
struct MarchTrace {
  uint32_t t[4], rounds, whole, hw_id, xcc_id;
};
__device__ MarchTrace g_march_trace[32768];  // [0, 16384): k_raymarch<0>, [16384, 32768): k_depth_peels (entry and exit only)
extern "C" int rgbdr_debug_march_trace(void* dst, size_t bytes)
{
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_march_trace), bytes, 0, hipMemcpyDeviceToHost);
}
#define MARCH_STAMP(k) trace_t[k] = (uint32_t)wall_clock64()
#else
#define MARCH_STAMP(k)
#endif

template <int MODE, int AHEAD = 1>
__global__ __launch_bounds__(256) void k_raymarch(RaymarchParams p)
{
  extern __shared__ unsigned s_empty[];  // MODE 0 with p.empty_bits: the workgroup's copy of the bitmap
#ifdef RGBDR_TRACE_BLOCKS
  uint32_t trace_t[4], trace_rounds = 0, trace_whole = 0;
  MARCH_STAMP(0);
#endif
  int px, py;
  wave_square_pixel(px, py);
  const bool inside = px < p.width && py < p.height;
  const size_t o = inside ? (size_t)py * p.width + px : 0;
  const float limit = p.limit;
  float4 rgba = make_float4(0.0f, 1.0f, 0.0f, 0.0f);  // ViewLod::enable clear colour
  float fdepth = 1.0f, fsamples = 0.0f;
  int khit = kNoHit;
  Ray r;
  r.covered = false;
  r.max_num = 0u;
  if (inside) r = ray_setup(p, px, py, o);
  const unsigned* empty_bits = nullptr;
  if (MODE == 0 && p.empty_bits) {  // (uniform) most workgroups of a frame have no ray to march: they skip the copy
    if (__syncthreads_or(r.covered && r.max_num > 0u)) {
      for (int w = threadIdx.y * 16 + threadIdx.x; w < p.empty_words; w += 256) s_empty[w] = p.empty_bits[w];
      __syncthreads();
      empty_bits = s_empty;
    }
  }
  if (MODE == 0) {
    // A wavefront lasts as long as its longest lane, a round of AHEAD samples is ~3 us of dependent instructions for a lane,
    // and a frame has a handful of rays that march 200-300 samples next to neighbours that take ten
    // (profiles/march_samples_probe.py): the frame would wait 0.1 ms for them.  So the wavefront's lanes march on their own,
    // round by round, while that is the shorter way; once a few long rays are all that is left, the whole wavefront
    // takes them one after the other, 64 samples a round.  (No lane has left the kernel: all 64 take part.)
    const int lane = (threadIdx.y * 16 + threadIdx.x) & 63;
    const bool marching = inside && r.covered;
    float sp[3] = {r.sp[0], r.sp[1], r.sp[2]};
    float prev = -limit;
    unsigned num = 0;
    bool hit = false;
    MARCH_STAMP(1);
    for (int round = 0;; ++round) {
      const bool live = marching && !hit && num < r.max_num;
      unsigned long long todo = __ballot(live);
      if (!todo) break;
      bool whole_wave = p.whole_wave == 1 && round >= 1;
      if (p.whole_wave == 0 && round >= 2 && __popcll(todo) <= 8) {  // rounds left either way (64 samples of the wavefront take about as long as 8 of a lane)
        const unsigned left = r.max_num - num;
        unsigned longest = 0, summed = 0;
        for (unsigned long long m = todo; m; m &= m - 1) {
          const unsigned l = (unsigned)__builtin_amdgcn_readlane((int)left, __ffsll((long long)m) - 1);
          longest = l > longest ? l : longest;
          summed += (l + 63u) >> 6;
        }
        whole_wave = summed < (longest + 7u) / 8u;
      }
      if (whole_wave) {
#ifdef RGBDR_TRACE_BLOCKS
        trace_whole = (uint32_t)__popcll(todo);
#endif
        for (; todo; todo &= todo - 1) march_whole_wave(p, r, __ffsll((long long)todo) - 1, lane, sp, prev, num, hit);
        break;
      }
      if (live) march_ahead<AHEAD>(p, r, sp, prev, num, hit, empty_bits, 1);
#ifdef RGBDR_TRACE_BLOCKS
      trace_rounds = (uint32_t)round + 1;
#endif
    }
    MARCH_STAMP(2);
    if (marching) {
      fsamples = (float)num * 0.0027f;
      if (hit) shade_fragment(p, sp, rgba, fdepth);
    }
#ifdef RGBDR_TRACE_BLOCKS
    MARCH_STAMP(3);
    if (lane == 0) {
      MarchTrace& t = g_march_trace[((blockIdx.y * gridDim.x + blockIdx.x) * 4 + ((threadIdx.y * 16 + threadIdx.x) >> 6)) % 16384];
      for (int k = 0; k < 4; ++k) t.t[k] = trace_t[k];
      t.rounds = trace_rounds;
      t.whole = trace_whole;
      t.hw_id = __builtin_amdgcn_s_getreg((31 << 11) | 4);
      t.xcc_id = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    }
#endif
    if (!inside) return;
  } else if (!inside) {
    return;
  } else if (r.covered) {
    float sp[3] = {r.sp[0], r.sp[1], r.sp[2]};
    if (MODE == 1) {
      // as in march_ahead: AHEAD samples in flight, looked at in order (a sample this slab does not own is fetched from the
      // nearest resident rows and not looked at)
      for (unsigned k0 = 0; k0 < r.max_num && khit == kNoHit; k0 += AHEAD) {
        float pos[AHEAD][3], dens[AHEAD];
#pragma unroll
        for (int j = 0; j < AHEAD; ++j) {
#pragma unroll
          for (int a = 0; a < 3; ++a) pos[j][a] = j == 0 ? sp[a] : pos[j - 1][a] + r.step[a];
          dens[j] = tsdf_sample(p, pos[j][0], pos[j][1], pos[j][2]);
        }
#pragma unroll
        for (int j = 0; j < AHEAD; ++j) {
          if (khit != kNoHit || !(k0 + j < r.max_num)) break;
          if (sample_owned(p, pos[j][2]) && dens[j] > 0.0f) khit = (int)(k0 + j);
#pragma unroll
          for (int a = 0; a < 3; ++a) sp[a] = pos[j][a] + r.step[a];
        }
      }
    } else {
      const int kmin = p.khit[o];
      if (kmin == kNoHit) {
        fsamples = (float)r.max_num * 0.0027f;
      } else {
        fsamples = (float)((unsigned)kmin + 1u) * 0.0027f;
        float prev_sp[3] = {sp[0], sp[1], sp[2]};
        for (int k = 0; k < kmin; ++k) {  // the same accumulation the single-volume march performs
#pragma unroll
          for (int a = 0; a < 3; ++a) {
            prev_sp[a] = sp[a];
            sp[a] += r.step[a];
          }
        }
        if (sample_owned(p, sp[2])) {
          khit = kmin;
          const float prev = kmin == 0 ? -limit : tsdf_sample(p, prev_sp[0], prev_sp[1], prev_sp[2]);
          const float density = tsdf_sample(p, sp[0], sp[1], sp[2]);
          const float f = prev / (density - prev);
#pragma unroll
          for (int a = 0; a < 3; ++a) sp[a] = (sp[a] - r.step[a]) - r.step[a] * f;
          shade_fragment(p, sp, rgba, fdepth);
        }
      }
    }
  }
  if (MODE == 1) {
    p.khit[o] = khit;
    return;
  }
  if (MODE == 2) p.khit[o] = khit;  // kNoHit unless this slab shaded the pixel
  p.out_color[o] = rgba;
  p.out_depth[o] = fdepth;
  p.out_samples[o] = fsamples;
}

// ---------------------------------------------------------------------------
// Brick depth peels: ReconIntegration::drawDepthLimits with
// glsl/bricks.{vs,gs,fs} -- instanced unit cubes of the occupied bricks, MIN blending
// of (z, -z, front ? 1 : z) into a target cleared to (1,0,1,0), faces towards a
// neighbour with counter > 10 dropped by the geometry shader.  Per pixel: walk the
// brick grid along the ray (Amanatides-Woo); entering an occupied brick is a front
// face, leaving one a back face; faces outside the depth range are clipped.
// Across a face on the grid's boundary the shader's "neighbour" is the brick its uint index
// arithmetic wraps to (bricks.gs:28-43, inc_bricks.glsl:25-27): the linear id one before / after
// (x), res.x before / after (y), res.x * res.y before / after (z) -- an aliased brick of the buffer
// unless that id falls before the first or past the last brick (then: not occupied).
__device__ __forceinline__ float peel_z(const PeelParams& p, const float* o, const float* d, float t)
{
  const float4 c = mat4_mul(p.pmv, o[0] + d[0] * t, o[1] + d[1] * t, o[2] + d[2] * t, 1.0f);
  return (c.z / c.w) * 0.5f + 0.5f;
}
__device__ __forceinline__ bool peel_in_grid(const PeelParams& p, const int* c)
{
  return !(c[0] < 0 || c[1] < 0 || c[2] < 0 || c[0] >= p.res_bricks[0] || c[1] >= p.res_bricks[1] || c[2] >= p.res_bricks[2]);
}
// (the same by linear id: the id is linear in the cell's coordinates, one cell outside the grid included)
// (k_peel_near keeps "counter > 10" as bit 2 of the brick's byte, so the walk needs no second array)
__device__ __forceinline__ bool peel_gt10_id(const PeelParams& p, long long id, long long nb) { return id >= 0 && id < nb && (p.cells[id] & 4u) != 0; }
__device__ __forceinline__ bool peel_gt10(const PeelParams& p, const int* c)
{
  // c may lie one cell outside the grid along one axis (see above): linear id with the shader's wrap-around
  const long long nb = (long long)p.res_bricks[0] * p.res_bricks[1] * p.res_bricks[2];
  const long long id = ((long long)c[2] * p.res_bricks[1] + c[1]) * p.res_bricks[0] + c[0];
  return id >= 0 && id < nb && p.counters[id] > 10u;
}

// Empty-space flags of the walk below.  A super-cell of 4 x 4 x 4 bricks is NEAR when a listed brick lies in it or next
// to it (the super-cell dilated by one brick, clipped to the grid).  While the walk is in a super-cell that is not near,
// neither the cell it is in, nor the one it came from, nor the one it goes to is listed, so no face can be crossed.
// One wavefront per super-cell finds the flag and writes a byte per brick of the super-cell: bit 0 = the brick is listed
// (the library's mask), bit 1 = its super-cell is near, bit 2 = its counter is above 10 (the geometry shader's cull of a
// face towards such a neighbour, bricks.gs:28-43) -- so the walk gets all three from the one byte it loads per cell.
// Most of a frame is empty space (3 % of the bricks are occupied in SURVEY 8d's scene, 13 % of the super-cells near).
__global__ __launch_bounds__(256) void k_peel_near(const uint8_t* mask, const uint32_t* counters, int rx, int ry, int rz, int sx, int sy, int sz,
                                                   uint8_t* cells, int force, unsigned near_blocks, PeelParams::EmptyTiles et)
{
  if (blockIdx.x >= near_blocks) {  // (k_empty_tiles' work: one dependent launch less in front of the march)
    empty_tiles_block(et.tile_state, et.epoch, et.TX, et.TY, et.TZ, (unsigned long long*)et.bits, blockIdx.x - near_blocks, threadIdx.x);
    return;
  }
  const int id = (int)(blockIdx.x * 4u + (threadIdx.x >> 6));  // a wavefront per super-cell
  if (id >= sx * sy * sz) return;
  const int lane = threadIdx.x & 63;
  const int cx = id % sx, cy = (id / sx) % sy, cz = id / (sx * sy);
  const int x0 = max(4 * cx - 1, 0), x1 = min(4 * cx + 4, rx - 1);
  const int y0 = max(4 * cy - 1, 0), y1 = min(4 * cy + 4, ry - 1);
  const int z0 = max(4 * cz - 1, 0), z1 = min(4 * cz + 4, rz - 1);
  const int nx = x1 - x0 + 1, ny = y1 - y0 + 1, n = nx * ny * (z1 - z0 + 1);
  unsigned any = 0;
  for (int k = lane; k < n; k += 64) {
    const int x = x0 + k % nx, y = y0 + (k / nx) % ny, z = z0 + k / (nx * ny);
    any |= mask[((size_t)z * ry + y) * rx + x];
  }
  const unsigned nearbit = (__ballot(any != 0) != 0 || force) ? 2u : 0u;
  const int bx = 4 * cx + (lane & 3), by = 4 * cy + ((lane >> 2) & 3), bz = 4 * cz + (lane >> 4);
  if (bx < rx && by < ry && bz < rz) {
    const size_t at = ((size_t)bz * ry + by) * rx + bx;
    cells[at] = (uint8_t)((mask[at] != 0 ? 1u : 0u) | nearbit | (counters[at] > 10u ? 4u : 0u));
  }
}

__global__ __launch_bounds__(256) void k_depth_peels(PeelParams p)
{
#ifdef RGBDR_TRACE_BLOCKS
  const uint32_t trace_t0 = (uint32_t)wall_clock64();
  struct PeelStamp {
    uint32_t t0;
    __device__ ~PeelStamp()
    {
      const int t = threadIdx.y * 16 + threadIdx.x;
      if (t & 63) return;
      MarchTrace& m = g_march_trace[16384 + ((blockIdx.y * gridDim.x + blockIdx.x) * 4 + (t >> 6)) % 16384];
      m.t[0] = t0;
      m.t[3] = (uint32_t)wall_clock64();
    }
  } peel_stamp{trace_t0};
#endif
  int px, py;
  wave_square_pixel(px, py);
  if (px >= p.width || py >= p.height) return;
  float r = 1.0f, gneg = 0.0f, b = 1.0f;
  do {
    const float4 o4 = mat4_mul(p.modelview_inv, 0.0f, 0.0f, 0.0f, 1.0f);
    const float4 pc = mat4_mul(p.img_to_eye, (float)px + 0.5f, (float)py + 0.5f, 1.0f, 1.0f);
    const float4 f4 = mat4_mul(p.modelview_inv, pc.x / pc.w, pc.y / pc.w, pc.z / pc.w, 1.0f);
    const float o[3] = {o4.x, o4.y, o4.z};
    const float d[3] = {f4.x - o4.x, f4.y - o4.y, f4.z - o4.z};
    float t0 = 0.0f, t1 = 1.0f;
    int entry_axis = -1;                    // the box face the ray enters the grid through, if it starts outside
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float lo = p.bbox_min[a], hi = p.bbox_min[a] + p.brick_size * (float)p.res_bricks[a];
      const float inv = 1.0f / d[a];
      const float ta = (lo - o[a]) * inv, tb = (hi - o[a]) * inv;
      const float tin = fminf(ta, tb);
      if (tin > t0) {
        t0 = tin;
        entry_axis = a;
      }
      t1 = fminf(t1, fmaxf(ta, tb));
    }
    if (!(t0 < t1)) break;
    int cell[3], stepi[3], prev[3];
    float tmax[3], tdelta[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float pos = (o[a] + d[a] * t0 - p.bbox_min[a]) / p.brick_size;
      int c = (int)floorf(pos);
      c = clampi(c, 0, p.res_bricks[a] - 1);
      cell[a] = c;
      if (d[a] > 0.0f) {
        stepi[a] = 1;
        tmax[a] = (p.bbox_min[a] + p.brick_size * (float)(c + 1) - o[a]) / d[a];
        tdelta[a] = p.brick_size / d[a];
      } else if (d[a] < 0.0f) {
        stepi[a] = -1;
        tmax[a] = (p.bbox_min[a] + p.brick_size * (float)c - o[a]) / d[a];
        tdelta[a] = -p.brick_size / d[a];
      } else {
        stepi[a] = 0;
        tmax[a] = __builtin_inff();
        tdelta[a] = __builtin_inff();
      }
    }
    // the cell outside the grid the ray comes from (its aliased brick decides the cull of the first front face)
    prev[0] = cell[0] - (entry_axis == 0 ? stepi[0] : 0);
    prev[1] = cell[1] - (entry_axis == 1 ? stepi[1] : 0);
    prev[2] = cell[2] - (entry_axis == 2 ? stepi[2] : 0);
    bool prev_in_grid = false, first = true;
    // One mask byte per cell, loaded one cell ahead: the walk itself needs no loads, so the next cell is known before
    // this cell's faces are looked at and its byte is in flight meanwhile.
    // Empty space (k_peel_near): a lane whose cell lies in a super-cell that neither holds nor touches a listed brick
    // crosses no face until it leaves that super-cell, so it takes the whole super-cell in ONE step.  The t values of the
    // three axes are independent sequences (t += delta, one addition per crossing), so the state the cell-by-cell walk
    // would have on leaving the super-cell follows without walking: per axis the values of its next (up to four) crossings,
    // the axis whose crossing of the super-cell's boundary comes first in the walk's order (smallest value, lowest axis
    // on a tie) is the exit, and of the other axes exactly the crossings that precede it in that order have happened.
    // Same additions in the same order per axis: the state is bit-identical, ~12 instead of ~65 instructions per cell.
    // The ray's first stretch -- from where it enters the grid to the first super-cell that holds or touches a listed
    // brick -- is taken a super-cell per step (k_peel_near explains why that is exact); from there on the walk goes cell
    // by cell.  (Alternating between the two for the rest of the ray -- as one loop, as loops nested in an outer one, or
    // written out A B A B A B -- executed fewer instructions but measured slower: profiles/r05_notes.md.)
    unsigned cur = p.cells[((size_t)cell[2] * p.res_bricks[1] + cell[1]) * p.res_bricks[0] + cell[0]];
    float tcur = t0;
    int steps = 0;                          // index of the current cell along the ray (cells 0 ... 4095 are looked at)
    bool ended = false;
    while ((cur & 2u) == 0) {
#define RGBDR_PEEL_AXIS(A)                                                                                   \
  const int n##A = stepi[A] > 0 ? 4 - (cell[A] & 3) : (cell[A] & 3) + 1; /* cells to the super-cell's boundary */ \
  const float v##A##0 = tmax[A], v##A##1 = v##A##0 + tdelta[A], v##A##2 = v##A##1 + tdelta[A], v##A##3 = v##A##2 + tdelta[A]; \
  const float T##A = n##A == 1 ? v##A##0 : n##A == 2 ? v##A##1 : n##A == 3 ? v##A##2 : v##A##3;
      RGBDR_PEEL_AXIS(0)
      RGBDR_PEEL_AXIS(1)
      RGBDR_PEEL_AXIS(2)
#undef RGBDR_PEEL_AXIS
      int ax = 0;
      float T = T0;
      if (T1 < T) {
        ax = 1;
        T = T1;
      }
      if (T2 < T) {
        ax = 2;
        T = T2;
      }
      if (!(T < t1) || steps >= 4096) {      // the ray ends inside this super-cell: no face
        ended = true;
        break;
      }
      // crossings of axis A that the walk makes before (T, ax): value below T, or equal to it on a lower axis;
      // the exit axis makes its n crossings
#define RGBDR_PEEL_TAKE(A)                                                                                    \
  {                                                                                                           \
    const bool tie = A < ax;                                                                                  \
    const int c = ax == A ? n##A                                                                              \
                          : (int)(v##A##0 < T || (tie && v##A##0 == T)) + (int)(v##A##1 < T || (tie && v##A##1 == T)) + \
                                (int)(v##A##2 < T || (tie && v##A##2 == T));                                  \
    const float after = c == 0 ? v##A##0 : c == 1 ? v##A##1 : c == 2 ? v##A##2 : c == 3 ? v##A##3 : v##A##3 + tdelta[A]; \
    tmax[A] = after;                                                                                          \
    cell[A] += c * stepi[A];                                                                                  \
    prev[A] = ax == A ? cell[A] - stepi[A] : cell[A];                                                         \
    steps += c;                                                                                               \
  }
      RGBDR_PEEL_TAKE(0)
      RGBDR_PEEL_TAKE(1)
      RGBDR_PEEL_TAKE(2)
#undef RGBDR_PEEL_TAKE
      tcur = T;
      first = false;
      prev_in_grid = true;
      if (!peel_in_grid(p, cell)) {          // out of the grid from an unlisted cell: no face
        ended = true;
        break;
      }
      cur = p.cells[((size_t)cell[2] * p.res_bricks[1] + cell[1]) * p.res_bricks[0] + cell[0]];
    }
    if (ended) break;
    // one byte per cell, loaded one cell ahead: the walk itself needs no loads, so the next cell is known before
    // this cell's faces are looked at and its byte is in flight meanwhile
    bool last_list = false;                 // was the cell the ray just left on the list?
    bool cur_list = (cur & 1u) != 0;
    bool cur_gt10 = (cur & 4u) != 0, prev_gt10 = false, prev_known = false;   // (the byte of the cell before the first is not at hand)
    // linear ids of the cell and of the one before it, advanced by the stride of the axis stepped along
    const long long rx = p.res_bricks[0], rxy = rx * p.res_bricks[1], nb = rxy * p.res_bricks[2];
    const long long stride0 = stepi[0], stride1 = stepi[1] * rx, stride2 = stepi[2] * rxy;
    long long idx = ((long long)cell[2] * p.res_bricks[1] + cell[1]) * rx + cell[0];
    long long pidx = ((long long)prev[2] * p.res_bricks[1] + prev[1]) * rx + prev[0];
    for (int iter = steps; iter < 4096; ++iter) {
      // the axis crossed next: the smallest value, the lowest axis on a tie (min3 + two equality tests)
      const float tnext = fminf(fminf(tmax[0], tmax[1]), tmax[2]);
      const bool a0 = tmax[0] == tnext, a1 = !a0 && tmax[1] == tnext, a2 = !(a0 || a1);
      const bool leaving = !(tnext < t1);
      const int n0 = cell[0] + (a0 ? stepi[0] : 0), n1 = cell[1] + (a1 ? stepi[1] : 0), n2 = cell[2] + (a2 ? stepi[2] : 0);
      const long long nidx = idx + (a0 ? stride0 : a1 ? stride1 : stride2);
      const bool next_in_grid = !leaving && (unsigned)n0 < (unsigned)p.res_bricks[0] && (unsigned)n1 < (unsigned)p.res_bricks[1] &&
                                (unsigned)n2 < (unsigned)p.res_bricks[2];
      const unsigned next = next_in_grid ? p.cells[nidx] : 0u;
      const bool next_list = (next & 1u) != 0;
      const bool prev_list = prev_in_grid && last_list;
      // most cells of a near super-cell are not listed themselves: one test keeps them out of the face logic
      if (cur_list || prev_list) {
        if (!(first && !(t0 > 0.0f))) {
          const float z = peel_z(p, o, d, tcur);
          if (z >= 0.0f && z <= 1.0f) {
            if (cur_list && !(prev_known ? prev_gt10 : peel_gt10_id(p, pidx, nb))) {
              r = fminf(r, z);
              gneg = fminf(gneg, -z);
            }
            if (prev_list && !cur_gt10) {
              r = fminf(r, z);
              gneg = fminf(gneg, -z);
              b = fminf(b, z);
            }
          }
        }
        if (leaving && cur_list && !peel_gt10_id(p, nidx, nb)) {
          const float z = peel_z(p, o, d, t1);
          if (t1 < 1.0f && z >= 0.0f && z <= 1.0f) {
            r = fminf(r, z);
            gneg = fminf(gneg, -z);
            b = fminf(b, z);
          }
        }
      }
      first = false;
      if (leaving) break;
      prev_in_grid = true;
      last_list = cur_list;
      prev_gt10 = cur_gt10;
      prev_known = true;
      cur_gt10 = (next & 4u) != 0;
      pidx = idx;
      idx = nidx;
      cell[0] = n0;
      cell[1] = n1;
      cell[2] = n2;
      tmax[0] = a0 ? tmax[0] + tdelta[0] : tmax[0];
      tmax[1] = a1 ? tmax[1] + tdelta[1] : tmax[1];
      tmax[2] = a2 ? tmax[2] + tdelta[2] : tmax[2];
      tcur = tnext;
      if (!next_in_grid) {
        if (last_list && !peel_gt10_id(p, idx, nb)) {
          const float z = peel_z(p, o, d, tcur);
          if (z >= 0.0f && z <= 1.0f) {
            r = fminf(r, z);
            gneg = fminf(gneg, -z);
            b = fminf(b, z);
          }
        }
        break;
      }
      cur_list = next_list;
    }
  } while (false);
  p.out[(size_t)py * p.width + px] = make_float4(r, gneg, b, 0.0f);
}

void launch_depth_peels(const PeelParams& p, hipStream_t s)
{
  // RGBDR_PEEL_ALLNEAR=1 (diagnostic, profiles/pmc_peels.sh): every super-cell counts as near, i.e. the plain walk
  static const int all_near = std::getenv("RGBDR_PEEL_ALLNEAR") ? 1 : 0;
  const int ns = p.res_super[0] * p.res_super[1] * p.res_super[2];
  const unsigned near_blocks = ((unsigned)ns + 3u) / 4u;
  const unsigned et_blocks = p.empty_tiles.bits ? ((unsigned)(p.empty_tiles.TX * p.empty_tiles.TY * p.empty_tiles.TZ) + 255u) / 256u : 0u;
  hipLaunchKernelGGL(k_peel_near, dim3(near_blocks + et_blocks), dim3(256), 0, s, p.mask, p.counters, p.res_bricks[0], p.res_bricks[1],
                     p.res_bricks[2], p.res_super[0], p.res_super[1], p.res_super[2], p.cells, all_near, near_blocks, p.empty_tiles);
  dim3 grid((p.width + 15) / 16, (p.height + 15) / 16);
  hipLaunchKernelGGL(k_depth_peels, grid, dim3(16, 16), 0, s, p);
}

void launch_raymarch(const RaymarchParams& p, int mode, hipStream_t s)
{
  dim3 grid((p.width + 15) / 16, (p.height + 15) / 16);
  const size_t lds = mode == 0 && p.empty_bits ? (size_t)p.empty_words * 4 : 0;
  if (mode == 0)
    if (p.skip_space)
      hipLaunchKernelGGL((k_raymarch<0, RGBDR_AHEAD_SKIP>), grid, dim3(16, 16), lds, s, p);
    else
      hipLaunchKernelGGL((k_raymarch<0, RGBDR_AHEAD_FULL>), grid, dim3(16, 16), lds, s, p);
  else if (mode == 1)
    if (p.skip_space)
      hipLaunchKernelGGL((k_raymarch<1, RGBDR_AHEAD_SKIP>), grid, dim3(16, 16), 0, s, p);
    else
      hipLaunchKernelGGL((k_raymarch<1, RGBDR_AHEAD_FULL>), grid, dim3(16, 16), 0, s, p);
  else
    hipLaunchKernelGGL(k_raymarch<2>, grid, dim3(16, 16), 0, s, p);
}

}  // namespace rgbdr
