// dxt.cuh -- one texel of a DXT1 / DXT5 colour frame straight from its blocks (squish::DecompressColour,
// external/squish/colourblock.cpp:140-214, with squish's integer arithmetic): what k_decode_dxt stores at (x, y).  Shared by
// the bilinear colour lookups of k_pre_depth and of the ray-marcher's shading, which read the frame as it was uploaded.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rgbdr {

// The colour half of one 4 x 4 block as two words -- x: the two 565 end points, y: sixteen 2-bit indices, row by
// row -- and one texel of it: what k_decode_dxt stores at (x, y), with squish's integer arithmetic (dxt_palette).
// k_pre_depth's bilinear lookup fetches each block it touches once (one 8-byte load; the four taps lie in one
// block for 9 of 16 positions) instead of six byte loads per tap.
__device__ __forceinline__ uint2 dxt_block(const uint8_t* __restrict__ layer, int bw, int mode, int bx, int by)
{
  return *reinterpret_cast<const uint2*>(layer + (size_t)(by * bw + bx) * (mode == 1 ? 8 : 16) + (mode == 1 ? 0 : 8));
}
__device__ __forceinline__ void dxt_texel(uint2 blk, int mode, int x, int y, int* rgb)
{
  const int a = (int)(blk.x & 0xffffu), bb = (int)(blk.x >> 16);
  const int idx = (int)(blk.y >> (8 * (y & 3) + 2 * (x & 3))) & 3;
  const bool three = mode == 1 && a <= bb;  // DXT1 block with a transparent fourth colour
  const int ca[3] = {(a >> 11) & 0x1f, (a >> 5) & 0x3f, a & 0x1f}, cb[3] = {(bb >> 11) & 0x1f, (bb >> 5) & 0x3f, bb & 0x1f};
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int c = i == 1 ? ((ca[i] << 2) | (ca[i] >> 4)) : ((ca[i] << 3) | (ca[i] >> 2));
    const int d = i == 1 ? ((cb[i] << 2) | (cb[i] >> 4)) : ((cb[i] << 3) | (cb[i] >> 2));
    const int m2 = three ? (c + d) / 2 : (2 * c + d) / 3, m3 = three ? 0 : (c + 2 * d) / 3;
    rgb[i] = idx == 0 ? c : (idx == 1 ? d : (idx == 2 ? m2 : m3));
  }
}

}  // namespace rgbdr
