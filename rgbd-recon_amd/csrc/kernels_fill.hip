// kernels_fill.hip -- hole filling of the ray-marched frame:
// ReconIntegration::fillColors (framework/reconstruction/recon_integration.cpp:280-339)
// with glsl/framebuffer_transfer.fs, glsl/tsdf_inpaint.fs, glsl/tsdf_colorfill.fs over
// the LOD atlas of ViewLod (framework/rendering/view_lod.cpp:24-61).  Two atlases
// of 1.5*W x H texels: "native" accumulates the LODs, "squeezed" is the copy the
// reference makes with framebuffer_transfer.fs (fetch at pass_TexCoord * full
// resolution = x squeezed by 2/3), which tsdf_inpaint.fs then reads.  Screen-space,
// one thread per output texel; a handful of small launches per frame.
#include <hip/hip_runtime.h>

#include "rgbdr_internal.hpp"
#include "sampling.cuh"

namespace rgbdr {

__device__ __forceinline__ void fc_fetch(const float4* __restrict__ col, const float* __restrict__ dep, int FW, int H,
                                         int x, int y, float4& c, float& d)
{
  if (x < 0 || y < 0 || x >= FW || y >= H) {  // texelFetch outside the texture
    c = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    d = 0.0f;
    return;
  }
  c = col[(size_t)y * FW + x];
  d = dep[(size_t)y * FW + x];
}

// cleared atlas (glClearColor(0,1,0,0), depth 1) with an optional W x H frame in LOD 0
__global__ void k_fc_init(FillLayout L, const float4* __restrict__ frame_col, const float* __restrict__ frame_dep,
                          float4* __restrict__ col, float* __restrict__ dep)
{
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= L.FW) return;
  float4 c = make_float4(0.0f, 1.0f, 0.0f, 0.0f);
  float d = 1.0f;
  if (frame_col && x < L.W) {
    c = frame_col[(size_t)y * L.W + x];
    d = frame_dep[(size_t)y * L.W + x];
  }
  col[(size_t)y * L.FW + x] = c;
  dep[(size_t)y * L.FW + x] = d;
}

// framebuffer_transfer.fs into the LOD-0 viewport of a freshly cleared atlas; texels [x0, x1) x [y0, ...) only
// (the reference redraws the whole atlas after every LOD; only the texels whose source lies in the LOD just
// inpainted can change, launch_fill_colors)
__global__ void k_fc_transfer(FillLayout L, const float4* __restrict__ scol, const float* __restrict__ sdep,
                              float4* __restrict__ dcol, float* __restrict__ ddep, int x0, int x1, int y0)
{
  const int x = x0 + (int)(blockIdx.x * blockDim.x + threadIdx.x), y = y0 + (int)blockIdx.y;
  if (x >= x1) return;
  float4 c = make_float4(0.0f, 1.0f, 0.0f, 0.0f);
  float d = 1.0f;
  if (x < L.W) {
    const float u = ((float)x + 0.5f) / (float)L.W, v = ((float)y + 0.5f) / (float)L.H;
    fc_fetch(scol, sdep, L.FW, L.H, (int)(u * (float)L.FW), (int)(v * (float)L.H), c, d);
  }
  dcol[(size_t)y * L.FW + x] = c;
  ddep[(size_t)y * L.FW + x] = d;
}

// tsdf_inpaint.fs: reads the squeezed atlas, writes LOD lod+1 of the native one
__global__ void k_fc_inpaint(FillLayout L, int lod, const float4* __restrict__ scol, const float* __restrict__ sdep,
                             float4* __restrict__ ncol, float* __restrict__ ndep)
{
  const int i = lod + 1;
  const int fx = blockIdx.x * blockDim.x + threadIdx.x, fy = blockIdx.y;
  if (fx >= L.res[i][0] || fy >= L.res[i][1]) return;
  const int gx = L.off[i][0] + fx, gy = L.off[i][1] + fy;  // gl_FragCoord (pixel_center_integer)
  const float tcx = ((float)gx - (float)L.off[i][0]) / (float)L.res[i][0];
  const float tcy = ((float)gy - (float)L.off[i][1]) / (float)L.res[i][1];
  const int lx = (int)((float)L.off[lod][0] + (float)L.res[lod][0] * tcx);
  const int ly = (int)((float)L.off[lod][1] + (float)L.res[lod][1] * tcy);
  const int pix = (int)((float)lx * (2.0f / 3.0f)), piy = (int)((float)ly * 1.0f);
  float sr[16], sg[16], sb[16], sdp[16];
  float depth_av = 0.0f;
  int num = 0;
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y) {
      float4 c;
      float d;
      fc_fetch(scol, sdep, L.FW, L.H, pix + x - 1, piy + y - 1, c, d);
      if (c.w <= 0.0f) {
        c.x = -1.0f;
      } else {
        depth_av += d;
        ++num;
      }
      sr[x + y * 4] = c.x;
      sg[x + y * 4] = c.y;
      sb[x + y * 4] = c.z;
      sdp[x + y * 4] = d;
    }
  float4 oc;
  float od;
  if (num == 0) {
    float4 c;
    fc_fetch(scol, sdep, L.FW, L.H, pix, piy, c, od);
    oc = od < 1.0f ? make_float4(0.0f, 0.0f, 0.0f, -1.0f) : make_float4(0.0f, 1.0f, 0.0f, 0.0f);
  } else {
    depth_av /= (float)num;
    float tr = 0.0f, tg = 0.0f, tb = 0.0f, td = 0.0f, tw = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; ++k)
      if (sr[k] >= 0.0f && sdp[k] >= depth_av) {
        tr += sr[k] * 1.0f;
        tg += sg[k] * 1.0f;
        tb += sb[k] * 1.0f;
        td += sdp[k] * 1.0f;
        tw += 1.0f;
      }
    oc = make_float4(tr / tw, tg / tw, tb / tw, 1.0f);
    od = td / tw;
  }
  ncol[(size_t)gy * L.FW + gx] = oc;
  ndep[(size_t)gy * L.FW + gx] = od;
}

__device__ __forceinline__ int fc_mirror(int i, int n)
{
  const int period = 2 * n;
  int k = i % period;
  if (k < 0) k += period;
  return k < n ? k : period - 1 - k;
}

// texture(texture_color, p): LINEAR + MIRRORED_REPEAT (view_lod.cpp:52-53)
__device__ __forceinline__ float4 fc_texture(const float4* __restrict__ col, int FW, int H, float u, float v)
{
  const float tx = u * (float)FW - 0.5f, ty = v * (float)H - 0.5f;
  const float fx = floorf(tx), fy = floorf(ty);
  const float ax = tx - fx, ay = ty - fy;
  const int jx = idx_from_floor(fx, FW * 4), jy = idx_from_floor(fy, H * 4);
  const int x0 = fc_mirror(jx, FW), x1 = fc_mirror(jx + 1, FW), y0 = fc_mirror(jy, H), y1 = fc_mirror(jy + 1, H);
  const float4 t00 = col[(size_t)y0 * FW + x0], t10 = col[(size_t)y0 * FW + x1];
  const float4 t01 = col[(size_t)y1 * FW + x0], t11 = col[(size_t)y1 * FW + x1];
  return make_float4(lerpf(lerpf(t00.x, t10.x, ax), lerpf(t01.x, t11.x, ax), ay), lerpf(lerpf(t00.y, t10.y, ax), lerpf(t01.y, t11.y, ax), ay),
                     lerpf(lerpf(t00.z, t10.z, ax), lerpf(t01.z, t11.z, ax), ay), lerpf(lerpf(t00.w, t10.w, ax), lerpf(t01.w, t11.w, ax), ay));
}

__device__ __forceinline__ float fc_clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

// tsdf_colorfill.fs into the W x H output
__global__ void k_fc_colorfill(FillLayout L, const float4* __restrict__ ncol, const float* __restrict__ ndep,
                               float4* __restrict__ out_col, float* __restrict__ out_dep)
{
  const int px = blockIdx.x * blockDim.x + threadIdx.x, py = blockIdx.y;
  if (px >= L.W) return;
  const float rix = 1.0f / (float)L.FW, riy = 1.0f / (float)L.H;
  const float tcx = (float)px / (float)L.res[0][0], tcy = (float)py / (float)L.res[0][1];
  const float ptx = ((float)px + 0.5f) / (float)L.W, pty = ((float)py + 0.5f) / (float)L.H;
  float4 c = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  float d;
  int level = 0;
  for (; level < L.num_lods; ++level) {
    const int cx = (int)((float)L.off[level][0] + (float)L.res[level][0] * tcx);
    const int cy = (int)((float)L.off[level][1] + (float)L.res[level][1] * tcy);
    fc_fetch(ncol, ndep, L.FW, L.H, cx, cy, c, d);
    if (c.w > 0.0f) break;
  }
  if (level > 0) {
    float p[2][2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int l = level + 1 + k;
      const float ox = l < 20 ? (float)L.off[l][0] : 0.0f, oy = l < 20 ? (float)L.off[l][1] : 0.0f;
      const float rx = l < 20 ? (float)L.res[l][0] : 0.0f, ry = l < 20 ? (float)L.res[l][1] : 0.0f;
      p[k][0] = fc_clampf(ox + rx * ptx, ox + 0.5f, (ox + rx) - 0.5f) * rix;
      p[k][1] = fc_clampf(oy + ry * pty, oy + 0.5f, (oy + ry) - 0.5f) * riy;
    }
    const float4 c1 = fc_texture(ncol, L.FW, L.H, p[0][0], p[0][1]);
    const float4 c2 = fc_texture(ncol, L.FW, L.H, p[1][0], p[1][1]);
    const float w1 = sqrtf(ptx * ptx + pty * pty);
    const float w2 = 1.0f - w1;
    c = make_float4((c1.x * w1 + c2.x * w2) / (w1 + w2), (c1.y * w1 + c2.y * w2) / (w1 + w2), (c1.z * w1 + c2.z * w2) / (w1 + w2),
                    (c1.w * w1 + c2.w * w2) / (w1 + w2));
  }
  float4 c0;
  float d0;
  fc_fetch(ncol, ndep, L.FW, L.H, (int)((float)L.off[0][0] + (float)L.res[0][0] * tcx),
           (int)((float)L.off[0][1] + (float)L.res[0][1] * tcy), c0, d0);
  out_col[(size_t)py * L.W + px] = c;
  out_dep[(size_t)py * L.W + px] = d0;
}

void launch_fill_colors(const FillLayout& L, const float4* frame_col, const float* frame_dep, float4* ncol, float* ndep,
                        float4* scol, float* sdep, float4* out_col, float* out_dep, hipStream_t s)
{
  const dim3 full((L.FW + 127) / 128, L.H), blk(128);
  hipLaunchKernelGGL(k_fc_init, full, blk, 0, s, L, frame_col, frame_dep, ncol, ndep);
  hipLaunchKernelGGL(k_fc_transfer, full, blk, 0, s, L, ncol, ndep, scol, sdep, 0, L.FW, 0);
  for (int i = 1; i < L.num_lods; ++i) {
    const dim3 g((L.res[i][0] + 127) / 128, L.res[i][1]);
    hipLaunchKernelGGL(k_fc_inpaint, g, blk, 0, s, L, i - 1, scol, sdep, ncol, ndep);
    // the squeezed copy again, where it can have changed: texel (x, y) of it reads the native atlas at
    // ((int)((x + 0.5) / W * FW), y), i.e. about 1.5 x -- the texels whose source lies in LOD i (+- 2 texels for
    // the rounding; recomputing a texel whose source did not change writes the same value)
    const int sx0 = L.off[i][0], sx1 = L.off[i][0] + L.res[i][0];
    int x0 = (int)((double)sx0 * L.W / L.FW) - 2, x1 = (int)((double)sx1 * L.W / L.FW) + 3;
    int y0 = L.off[i][1] - 1, y1 = L.off[i][1] + L.res[i][1] + 1;
    x0 = x0 < 0 ? 0 : x0;
    x1 = x1 > L.W ? L.W : x1;
    y0 = y0 < 0 ? 0 : y0;
    y1 = y1 > L.H ? L.H : y1;
    if (x1 > x0 && y1 > y0)
      hipLaunchKernelGGL(k_fc_transfer, dim3((x1 - x0 + 127) / 128, y1 - y0), blk, 0, s, L, ncol, ndep, scol, sdep, x0, x1, y0);
  }
  hipLaunchKernelGGL(k_fc_colorfill, dim3((L.W + 127) / 128, L.H), blk, 0, s, L, ncol, ndep, out_col, out_dep);
}

}  // namespace rgbdr
