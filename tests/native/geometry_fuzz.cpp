// Host-side geometry of the library (csrc/geometry.cpp: no device calls) under AddressSanitizer + UBSan with random and
// hostile configurations: every float -> int conversion, table index and division has to stay defined whatever a caller
// puts into rgbdr_config (GPU sanitizers do not exist on the pool; this is the part of the product that runs on the host).
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include
//       tests/native/geometry_fuzz.cpp rgbd-recon_amd/csrc/geometry.cpp
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <random>
#include <string>

#include "../../rgbd-recon_amd/csrc/rgbdr_internal.hpp"

using namespace rgbdr;

int main(int argc, char** argv)
{
  const int iters = argc > 1 ? std::atoi(argv[1]) : 3000;
  std::mt19937 rng(1234);
  const float inf = std::numeric_limits<float>::infinity(), nan = std::numeric_limits<float>::quiet_NaN();
  const float special[] = {0.0f, -0.0f, 1e-30f, 1e-9f, 1e-4f, 0.007f, 0.01f, 0.0625f, 0.1f, 1.0f, 3.0f, 1e4f, 1e20f, 3e38f, inf, -inf, nan, -1.0f};
  auto pick = [&](float lo, float hi) {
    if (rng() % 4 == 0) return special[rng() % (sizeof special / sizeof special[0])];
    return lo + (hi - lo) * (float)(rng() % 100000) / 100000.0f;
  };
  long ok = 0, refused = 0, tables = 0;
  for (int it = 0; it < iters; ++it) {
    rgbdr_config c{};
    c.struct_size = sizeof c;
    c.num_sensors = 1 + (int)(rng() % 8);
    c.depth_w = 64, c.depth_h = 53, c.color_w = 64, c.color_h = 53;
    for (int a = 0; a < 3; ++a) {
      c.bbox_min[a] = pick(-2.0f, 0.0f);
      c.bbox_max[a] = pick(0.1f, 2.5f);
      c.res_override[a] = rng() % 5 == 0 ? (int32_t)(rng() % 3 == 0 ? rng() : rng() % 300) - (rng() % 7 == 0 ? 50 : 0) : 0;
    }
    c.voxel_size = pick(0.004f, 0.2f);
    c.brick_size = pick(0.004f, 0.5f);
    c.tsdf_limit = pick(0.001f, 0.3f);
    c.slab_count = rng() % 3 == 0 ? (int)(rng() % 12) - 1 : 1;
    c.slab_rank = c.slab_count > 1 ? (int)(rng() % (unsigned)(c.slab_count + 1)) - (rng() % 9 == 0) : 0;
    rgbdr_geometry g{};
    std::string err;
    const int rc = compute_geometry(c, &g, &err);
    if (rc != RGBDR_OK) {
      ++refused;
      continue;
    }
    ++ok;
    // what an accepted configuration promises
    for (int a = 0; a < 3; ++a)
      if (g.res_volume[a] < 1 || g.res_volume[a] > 32768 || g.tiles[a] != (g.res_volume[a] + 7) / 8 || g.res_bricks[a] < 1) {
        std::fprintf(stderr, "bad geometry accepted (axis %d: res %d tiles %d bricks %d)\n", a, g.res_volume[a], g.tiles[a], g.res_bricks[a]);
        return 1;
      }
    if (g.slab_tile_z0 < 0 || g.slab_tile_z1 <= g.slab_tile_z0 || g.slab_tile_z1 > g.tiles[2]) {
      std::fprintf(stderr, "bad slab range accepted\n");
      return 1;
    }
    // the brick -> voxel tables only for grids a context could hold (the tables are O(res) per axis)
    if ((long long)g.res_volume[0] * g.res_volume[1] * g.res_volume[2] <= (1ll << 27)) {
      BrickTables t;
      std::string e2;
      if (compute_brick_tables(c, g, &t, &e2) == RGBDR_OK) {
        ++tables;
        for (int a = 0; a < 3; ++a)
          if ((int)t.vox[a].size() != g.res_volume[a] || (int)t.tile[a].size() != g.tiles[a]) {
            std::fprintf(stderr, "brick tables of the wrong size\n");
            return 1;
          }
      }
    }
  }
  // slab ranges partition the tile layers
  for (int tiles = 1; tiles < 70; ++tiles)
    for (int count = -1; count <= tiles + 2; ++count)
      for (int rank = -1; rank <= count + 1; ++rank) {
        int t0 = -7, t1 = -7;
        (void)slab_range(tiles, count, rank, &t0, &t1);
      }
  // the LOD atlas of the hole filling for every small viewport and a few large ones
  for (int w = 1; w < 80; ++w)
    for (int h = 1; h < 80; h += 3) {
      FillLayout L;
      make_fill_layout(w, h, &L);
      if (L.num_lods < 1 || L.num_lods > 20) return 1;
    }
  for (int wh : {1280, 1920, 4096, 32768}) {
    FillLayout L;
    make_fill_layout(wh, (wh * 9) / 16, &L);
    if (L.num_lods < 1 || L.num_lods > 20) return 1;
  }
  // camera position / frustum planes from degenerate calibration volumes (zeros, NaN, a single cell)
  for (uint32_t r : {1u, 2u, 5u}) {
    const uint32_t res[3] = {r, r, r};
    std::vector<float> v((size_t)r * r * r * 3);
    for (int kind = 0; kind < 3; ++kind) {
      for (auto& x : v) x = kind == 0 ? 0.0f : (kind == 1 ? nan : (float)(rng() % 1000) / 100.0f);
      float cam[3], planes[6][4];
      camera_position(v.data(), res, cam);
      frustum_planes(v.data(), res, planes);
    }
  }
  for (int rz : {1, 2, 7, 128})
    for (int Z : {1, 8, 50, 512})
      for (int z0 = 0; z0 < Z; z0 += (Z + 3) / 4) {
        int lo, hi;
        lut_z_range(rz, Z, z0, Z, &lo, &hi);
        if (lo < 0 || hi >= rz || lo > hi) return 1;
      }
  std::printf("geometry fuzz: %ld configurations accepted, %ld refused, %ld brick tables built\n", ok, refused, tables);
  return 0;
}
