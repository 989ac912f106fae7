// fill_taps.cuh -- where the 4 x 4 taps of glsl/tsdf_inpaint.fs lie in the native atlas, as a function of the viewport
// size alone.  One definition for the host (make_fill_tables, geometry.cpp: the tables of the large and the LDS-resident
// passes) and the device (kernels_fill.hip: the passes in between compute them per texel, a cold table load costs more
// than the arithmetic), so both produce the same integers: plain IEEE binary32, built with -ffp-contract=off.
//
// tsdf_inpaint.fs:37-46: tex_coord = (gl_FragCoord.xy - offset[lod + 1]) / resolution[lod + 1];
// pos_int = ivec2(vec2(to_lod_pos(tex_coord, lod)) * vec2(2/3, 1)); taps at pos_int - 1 .. + 2 of the SQUEEZED atlas, whose
// texel (px, py) is native texel ivec2(pass_TexCoord * resolution_tex) with pass_TexCoord the texel's centre over the
// LOD-0 viewport (framebuffer_transfer.fs:13 drawn by fillColors, recon_integration.cpp:283-290); the rest of the squeezed
// atlas keeps the clear colour.
#pragma once
#include "rgbdr_internal.hpp"

#ifdef __HIPCC__
#define RGBDR_HD __host__ __device__
#else
#define RGBDR_HD  // tests/native builds geometry.cpp with g++ for the sanitizers
#endif

namespace rgbdr {

// N column of tap column t (0..3) for texel column fx of LOD i, or FC_TAP_OUTSIDE / FC_TAP_CLEAR
RGBDR_HD inline int fc_tap_column(const FillLayout& L, int i, int fx, int t)
{
  const int lod = i - 1;
  const int gx = L.off[i][0] + fx;  // gl_FragCoord.x (pixel_center_integer)
  const float tcx = ((float)gx - (float)L.off[i][0]) / (float)L.res[i][0];
  const int lx = (int)((float)L.off[lod][0] + (float)L.res[lod][0] * tcx);
  const int pix = (int)((float)lx * (2.0f / 3.0f));
  const int px = pix + t - 1;
  if (px < 0 || px >= L.FW) return FC_TAP_OUTSIDE;
  if (px >= L.W) return FC_TAP_CLEAR;
  const int e = (int)((((float)px + 0.5f) / (float)L.W) * (float)L.FW);
  return (e < 0 || e >= L.FW) ? FC_TAP_OUTSIDE : e;
}

// N row of tap row t for texel row fy of LOD i (| FC_ROW_CLEAR: under the LODs computed before pass i -- LOD i itself, the
// later ones, the gaps beside them -- where the band still holds the clear colour), or FC_TAP_OUTSIDE
RGBDR_HD inline int fc_tap_row(const FillLayout& L, int i, int fy, int t)
{
  const int lod = i - 1;
  const int gy = L.off[i][1] + fy;
  const float tcy = ((float)gy - (float)L.off[i][1]) / (float)L.res[i][1];
  const int ly = (int)((float)L.off[lod][1] + (float)L.res[lod][1] * tcy);
  const int piy = (int)((float)ly * 1.0f);
  const int py = piy + t - 1;
  if (py < 0 || py >= L.H) return FC_TAP_OUTSIDE;
  const int e = (int)((((float)py + 0.5f) / (float)L.H) * (float)L.H);
  if (e < 0 || e >= L.H) return FC_TAP_OUTSIDE;
  return (i == 1 || e < L.off[i - 1][1]) ? (e | FC_ROW_CLEAR) : e;
}

}  // namespace rgbdr
