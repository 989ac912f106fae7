// geometry.cpp -- host-only grid / brick / slab arithmetic.  No device calls, so
// the `not gpu` tests exercise it through the C ABI without a GPU.
//
// Restates (not copies) the reference's CPU-side geometry:
//   setVoxelSize   framework/reconstruction/recon_integration.cpp:341-354
//   setBrickSize   :474-484
//   divideBox      :361-388   (float-accumulating while loops -> m_res_bricks)
//   Frustum::getCameraPos  framework/calibration/frustum.cpp:21-33,97-111
#include <cmath>
#include <cstring>

#include "rgbdr_internal.hpp"
#include "fill_taps.cuh"

namespace rgbdr {

int slab_range(int tiles_z, int count, int rank, int* t0, int* t1)
{
  if (count <= 0) count = 1;
  if (tiles_z <= 0 || rank < 0 || rank >= count) return RGBDR_ERR_INVALID_ARGUMENT;
  // whole tile layers, the first (tiles_z % count) slabs get one layer more
  const int base = tiles_z / count, rem = tiles_z % count;
  *t0 = rank * base + (rank < rem ? rank : rem);
  *t1 = *t0 + base + (rank < rem ? 1 : 0);
  return RGBDR_OK;
}

int compute_geometry(const rgbdr_config& cfg, rgbdr_geometry* g, std::string* err)
{
  auto fail = [&](const char* m) {
    if (err) *err = m;
    return (int)RGBDR_ERR_INVALID_ARGUMENT;
  };
  // finite everywhere first: every conversion and loop below assumes it (an infinite box side makes divideBox's loop
  // condition NaN, an infinite brick a float -> int conversion of infinity; found by tests/native/geometry_fuzz.cpp)
  if (!(cfg.voxel_size > 0.0f) || !std::isfinite(cfg.voxel_size)) return fail("voxel_size must be a finite number > 0");
  if (!(cfg.brick_size > 0.0f) || !std::isfinite(cfg.brick_size)) return fail("brick_size must be a finite number > 0");
  for (int a = 0; a < 3; ++a) {
    if (!std::isfinite(cfg.bbox_min[a]) || !std::isfinite(cfg.bbox_max[a]) || !std::isfinite(cfg.bbox_max[a] - cfg.bbox_min[a]))
      return fail("the bounding box must be finite");
    if (!(cfg.bbox_max[a] > cfg.bbox_min[a])) return fail("bbox_max must exceed bbox_min on every axis");
  }

  for (int a = 0; a < 3; ++a) {
    // (the quotient is compared as a float: converting a value beyond INT_MAX, an infinity or a NaN is undefined)
    const float cells = cfg.res_override[a] > 0 ? (float)cfg.res_override[a] : std::ceil((cfg.bbox_max[a] - cfg.bbox_min[a]) / cfg.voxel_size);
    if (!(cells >= 1.0f)) return fail("empty volume");
    if (!(cells <= (float)kMaxRes)) return fail("more than 32768 voxels along an axis");
    g->res_volume[a] = (int)cells;
  }
  // tile indices are 31-bit (bit 31 of a work-list entry is the whole-tile flag, kernels_bricks.hip) and every size below
  // is computed from these counts: 2^31 tiles = 2^40 voxels = 4 TiB of TSDF, fourteen times an MI355X's HBM
  if ((long long)((g->res_volume[0] + kTile - 1) / kTile) * ((g->res_volume[1] + kTile - 1) / kTile) * ((g->res_volume[2] + kTile - 1) / kTile) >=
      (1ll << 31))
    return fail("volume of 2^31 or more 8x8x8 tiles");
  // setBrickSize: m_brick_size = m_voxel_size * round(size / m_voxel_size)
  float ratio = std::round(cfg.brick_size / cfg.voxel_size);
  if (ratio < 1.0f) ratio = 1.0f;
  if (!(ratio <= 1.0e6f)) return fail("brick_size of more than a million voxels");   // (and no float -> int conversion of a huge ratio)
  g->brick_size = cfg.voxel_size * ratio;
  g->brick_voxels = (int)ratio;
  for (int a = 0; a < 3; ++a) {
    g->brick_voxels_axis[a] = g->brick_voxels;
    if (cfg.res_override[a] > 0) {  // voxel edge on this axis = extent / res
      const float edge = (cfg.bbox_max[a] - cfg.bbox_min[a]) / (float)cfg.res_override[a];
      float r = std::round(g->brick_size / edge);
      if (!(r <= 1.0e6f)) return fail("brick_size of more than a million voxels");       // (also NaN)
      g->brick_voxels_axis[a] = r < 1.0f ? 1 : (int)r;
    }
  }
  // divideBox: bricks per axis from the reference's float-accumulating loop alone
  // (recon_integration.cpp:366-388) -- m_res_bricks, the divisor of occupiedRatio and the
  // clamp of mark_brick.  Voxels past the last brick belong to no brick (they stay -limit
  // in the brick-skipping sweep, as in the reference); which voxels a brick holds is
  // compute_brick_tables() below.
  for (int a = 0; a < 3; ++a) {
    const float mn = cfg.bbox_min[a];
    const float size = cfg.bbox_max[a] - mn;
    float start = mn;
    int n = 0;
    while (size - start + mn > 0.0f) {
      start += g->brick_size;
      ++n;
      if (n > 65535) return fail("brick grid too fine");
    }
    if (n < 1) return fail("the brick grid is empty along an axis");
    g->res_bricks[a] = n;
  }
  const long long nb = (long long)g->res_bricks[0] * g->res_bricks[1] * g->res_bricks[2];
  if (nb > (1ll << 30)) return fail("too many bricks");
  g->num_bricks = (int)nb;
  for (int a = 0; a < 3; ++a) g->tiles[a] = (g->res_volume[a] + kTile - 1) / kTile;
  int count = cfg.slab_count <= 0 ? 1 : cfg.slab_count;
  if (count > g->tiles[2]) return fail("more slabs than tile layers");
  int rc = slab_range(g->tiles[2], count, cfg.slab_rank, &g->slab_tile_z0, &g->slab_tile_z1);
  if (rc != RGBDR_OK) return fail("slab_rank out of range");
  g->slab_voxel_z0 = g->slab_tile_z0 * kTile;
  g->slab_voxel_z1 = g->slab_tile_z1 * kTile;
  if (g->slab_voxel_z1 > g->res_volume[2]) g->slab_voxel_z1 = g->res_volume[2];
  g->halo_tile_layers = 0;
  if (count > 1) {
    // a ray-marcher stepping limit/2 samples, refines and takes +-limit/2 gradients up to
    // limit * res_z + 2 voxel rows beyond the rows a slab owns (DESIGN.md "Multi-GPU")
    // (tsdf_limit is a fraction of the volume's unit cube: a limit at which the halo would span the whole grid -- or a
    // NaN -- cannot be served by any slab)
    const float frows = std::ceil(cfg.tsdf_limit * (float)g->res_volume[2]);
    if (!(frows >= 0.0f && frows <= (float)kMaxRes)) return fail("tsdf_limit must be a finite number in (0, 1]: the slab halo would exceed the grid");
    const int rows = (int)frows + 2;
    g->halo_tile_layers = rows <= kTile ? 1 : (rows + kTile - 1) / kTile;
    if (g->slab_tile_z1 - g->slab_tile_z0 < g->halo_tile_layers) return fail("slab thinner than its halo");
  }
  return RGBDR_OK;
}

// Which voxels a brick holds, exactly as divideBox + VolumeSampler::containedVoxels build the
// per-brick index lists (recon_integration.cpp:366-375, volume_sampler.cpp:50-62): per axis,
//   pos_n = (start - min) / size, size_n = min(brick_size, size - start + min) / size, step = 1 / res,
//   for (unsigned v = pos_n / step; v < (pos_n + size_n) / step; ++v)
// in binary32, `start` accumulated by += brick_size.  The float -> unsigned truncation and the
// float upper bound make neighbouring bricks share a voxel row wherever the quotients round up
// (at the reference's default 200 x 221 x 200 grid with 0.1 m bricks almost every brick also holds
// the first row of the next one), and the last brick can reach one index past the axis end.
int compute_brick_tables(const rgbdr_config& cfg, const rgbdr_geometry& g, BrickTables* t, std::string* err)
{
  for (int a = 0; a < 3; ++a) {
    const int res = g.res_volume[a];
    const float mn = cfg.bbox_min[a];
    const float size = cfg.bbox_max[a] - mn;
    const float step = 1.0f / (float)res;
    float start = mn;
    t->first[a].clear();
    t->last[a].clear();
    while (size - start + mn > 0.0f) {
      const float rest = size - start + mn;
      const float bsz = g.brick_size < rest ? g.brick_size : rest;  // glm::min(fvec3{m_brick_size}, size - start + min)
      const float pos_n = (start - mn) / size;
      const float size_n = bsz / size;
      const unsigned lo = (unsigned)(pos_n / step);
      const float bound = (pos_n + size_n) / step;
      long long hi = (long long)lo - 1;
      for (unsigned v = lo; (float)v < bound; ++v) {
        hi = v;
        if (v > lo + (1u << 22)) break;
      }
      t->first[a].push_back((int32_t)lo);
      t->last[a].push_back((int32_t)hi);
      start += g.brick_size;
      if ((int)t->first[a].size() > 65535) {
        if (err) *err = "brick grid too fine";
        return RGBDR_ERR_INVALID_ARGUMENT;
      }
    }
    const int nb = (int)t->first[a].size();
    t->vox[a].assign((size_t)res, 0x0000ffffu);  // lo = 0xffff, hi = 0: no brick
    t->overflow[a] = 0;
    for (int b = 0; b < nb; ++b) {
      const int lo = t->first[a][b], hi = t->last[a][b];
      if (hi >= res) {
        // indices past the axis end alias other voxels through the linear index
        // z*X*Y + y*X + x (integrate kernel: voxel_occupied); only the last brick can do that
        if (b != nb - 1) {
          if (err) *err = "a brick other than the last one reaches past the volume";
          return RGBDR_ERR_INVALID_ARGUMENT;
        }
        t->overflow[a] = hi - (res - 1);
      }
      for (int v = lo; v <= hi && v < res; ++v) {
        uint32_t e = t->vox[a][v];
        uint32_t l = e & 0xffffu, h = e >> 16;
        if (l > h) {  // first brick holding v
          l = (uint32_t)b;
          h = (uint32_t)b;
        } else {
          if ((uint32_t)b < l) l = (uint32_t)b;
          if ((uint32_t)b > h) h = (uint32_t)b;
        }
        t->vox[a][v] = l | (h << 16);
      }
    }
    const int ntile = (res + kTile - 1) / kTile;
    t->tile[a].assign((size_t)ntile, 0x0000ffffu);
    for (int v = 0; v < res; ++v) {
      const uint32_t e = t->vox[a][v];
      if ((e & 0xffffu) > (e >> 16)) continue;
      uint32_t& d = t->tile[a][v / kTile];
      uint32_t l = d & 0xffffu, h = d >> 16;
      if (l > h) {
        l = e & 0xffffu;
        h = e >> 16;
      } else {
        if ((e & 0xffffu) < l) l = e & 0xffffu;
        if ((e >> 16) > h) h = e >> 16;
      }
      d = l | (h << 16);
    }
    // a tile is "whole" along this axis when its 8 coordinates all lie inside the volume and all belong to
    // at least one brick: then "every brick the tile touches is occupied" implies "every voxel is"
    t->whole[a].assign((size_t)ntile, 0u);
    for (int tt = 0; tt < ntile; ++tt) {
      bool w = (tt + 1) * kTile <= res;
      for (int v = tt * kTile; w && v < (tt + 1) * kTile; ++v) w = (t->vox[a][v] & 0xffffu) <= (t->vox[a][v] >> 16);
      t->whole[a][tt] = w ? 1u : 0u;
    }
  }
  return RGBDR_OK;
}

static inline float dot3(const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

static void closest_point(const float* p, const float* u, const float* q, const float* v, float* o)
{
  const float w0[3] = {p[0] - q[0], p[1] - q[1], p[2] - q[2]};
  const float a = dot3(u, u), b = dot3(u, v), c = dot3(v, v), d = dot3(u, w0), e = dot3(v, w0);
  const float sc = (b * e - c * d) / (a * c - b * b);
  const float tc = (a * e - b * d) / (a * c - b * b);
  for (int k = 0; k < 3; ++k) o[k] = ((p[k] + u[k] * sc) + (q[k] + v[k] * tc)) * 0.5f;
}

void camera_position(const float* xyz, const uint32_t res[3], float out[3])
{
  const uint32_t ex = res[0] - 1, ey = res[1] - 1, ez = res[2] - 1;
  // corner order of getCornerPoints: near face (z = 0) then far face (z = end)
  const uint32_t cx[8] = {0, 0, ex, ex, 0, 0, ex, ex};
  const uint32_t cy[8] = {0, ey, ey, 0, 0, ey, ey, 0};
  const uint32_t cz[8] = {0, 0, 0, 0, ez, ez, ez, ez};
  float c[8][3];
  for (int i = 0; i < 8; ++i)
    std::memcpy(c[i], xyz + (((size_t)cz[i] * res[1] + cy[i]) * res[0] + cx[i]) * 3, 12);
  float cn[3], cf[3], view[3];
  for (int k = 0; k < 3; ++k) {
    cn[k] = (c[0][k] + c[1][k] + c[2][k] + c[3][k]) / 4.0f;
    cf[k] = (c[4][k] + c[5][k] + c[6][k] + c[7][k]) / 4.0f;
    view[k] = cf[k] - cn[k];
  }
  float pt[4][3];
  for (int i = 0; i < 4; ++i) {
    const float edge[3] = {c[i][0] - c[i + 4][0], c[i][1] - c[i + 4][1], c[i][2] - c[i + 4][2]};
    closest_point(c[i], edge, cn, view, pt[i]);
  }
  for (int k = 0; k < 3; ++k) out[k] = (pt[0][k] + pt[1][k] + pt[2][k] + pt[3][k]) / 4.0f;
}

void make_fill_layout(int W, int H, FillLayout* L)
{
  std::memset(L, 0, sizeof(*L));
  L->W = W;
  L->H = H;
  L->FW = (int)((float)W * 1.5f);  // m_resolution_full = uvec2{width * 1.5f, height}
  L->num_lods = 1 + (int)std::floor(std::log2((float)(W < H ? W : H)));
  if (L->num_lods > 20) L->num_lods = 20;  // the shaders' uniform arrays hold 20 entries
  int oy = H;
  for (int i = 0; i < L->num_lods; ++i) {
    L->res[i][0] = (int)std::floor((float)W / std::pow(2.0f, (float)i));
    L->res[i][1] = (int)std::floor((float)H / std::pow(2.0f, (float)i));
    if (i > 0) {
      oy -= L->res[i][1];
      L->off[i][0] = W;
      L->off[i][1] = oy;
    }
  }
}

// Tap tables of the inpaint passes (FillTabs): fill_taps.cuh evaluated for every texel column / row of every LOD >= 1.
void make_fill_tables(const FillLayout& L, std::vector<int>* xt, std::vector<int>* yt, FillTabs* T)
{
  xt->clear();
  yt->clear();
  std::memset(T->xbase, 0, sizeof(T->xbase));
  std::memset(T->ybase, 0, sizeof(T->ybase));
  for (int i = 1; i < L.num_lods; ++i) {
    T->xbase[i] = (int)(xt->size() / 4);
    T->ybase[i] = (int)(yt->size() / 4);
    for (int fx = 0; fx < L.res[i][0]; ++fx)
      for (int t = 0; t < 4; ++t) xt->push_back(fc_tap_column(L, i, fx, t));
    for (int fy = 0; fy < L.res[i][1]; ++fy)
      for (int t = 0; t < 4; ++t) yt->push_back(fc_tap_row(L, i, fy, t));
  }
  T->nx = (int)(xt->size() / 4);
  T->ny = (int)(yt->size() / 4);
}

// Does the sampled lattice of a cv_xyz volume fold?  The certificate of the device's inverse search (kernels_invert.hip) is
// argued for a lattice whose cells all have the same orientation: the triple product of a cell's three index directions
// keeps its sign.  A lens model that lets it change sign somewhere (a barrel distortion strong enough to turn the image
// back on itself) or a damaged file does not fit the argument; the search then scans exhaustively (still exact, slow).
// Volumes thinner than two samples along an axis have no cells: not folded.
bool lattice_folds(const float* xyz, const uint32_t res[3])
{
  if (res[0] < 2 || res[1] < 2 || res[2] < 2) return false;
  const size_t sx = 3, sy = (size_t)res[0] * 3, sz = (size_t)res[0] * res[1] * 3;
  bool pos = false, neg = false;
  for (uint32_t z = 0; z + 1 < res[2]; ++z)
    for (uint32_t y = 0; y + 1 < res[1]; ++y)
      for (uint32_t x = 0; x + 1 < res[0]; ++x) {
        const float* o = xyz + z * sz + y * sy + x * sx;
        double e[3][3];
        for (int k = 0; k < 3; ++k) {
          e[0][k] = (double)o[sx + k] - o[k];
          e[1][k] = (double)o[sy + k] - o[k];
          e[2][k] = (double)o[sz + k] - o[k];
        }
        const double det = e[0][0] * (e[1][1] * e[2][2] - e[1][2] * e[2][1]) - e[0][1] * (e[1][0] * e[2][2] - e[1][2] * e[2][0]) +
                           e[0][2] * (e[1][0] * e[2][1] - e[1][1] * e[2][0]);
        if (det > 0.0)
          pos = true;
        else if (det < 0.0)
          neg = true;
        else
          return true;  // a degenerate (or NaN) cell
        if (pos && neg) return true;
      }
  return false;
}

void frustum_planes(const float* xyz, const uint32_t res[3], float planes[6][4])
{
  const uint32_t ex = res[0] - 1, ey = res[1] - 1, ez = res[2] - 1;
  const uint32_t cx[8] = {0, 0, ex, ex, 0, 0, ex, ex};
  const uint32_t cy[8] = {0, ey, ey, 0, 0, ey, ey, 0};
  const uint32_t cz[8] = {0, 0, 0, 0, ez, ez, ez, ez};
  float c[8][3], e[12][3];
  for (int i = 0; i < 8; ++i)
    std::memcpy(c[i], xyz + (((size_t)cz[i] * res[1] + cy[i]) * res[0] + cx[i]) * 3, 12);
  // edge centres, side centres and side normals in the reference's numbering
  static const int ea[12] = {0, 1, 2, 3, 4, 5, 6, 7, 0, 1, 2, 3}, eb[12] = {1, 2, 3, 0, 5, 6, 7, 4, 4, 5, 6, 7};
  for (int i = 0; i < 12; ++i)
    for (int k = 0; k < 3; ++k) e[i][k] = (c[ea[i]][k] + c[eb[i]][k]) * 0.5f;
  static const int side[6][4] = {{0, 1, 2, 3}, {4, 5, 6, 7}, {0, 1, 4, 5}, {2, 3, 6, 7}, {1, 2, 5, 6}, {0, 3, 4, 7}};
  static const int nrm[6][4] = {{0, 2, 3, 2}, {4, 6, 5, 7}, {0, 4, 9, 8}, {2, 6, 11, 10}, {9, 10, 1, 5}, {8, 11, 7, 3}};
  for (int i = 0; i < 6; ++i) {
    float centre[3], a[3], b[3];
    for (int k = 0; k < 3; ++k) {
      centre[k] = (c[side[i][0]][k] + c[side[i][1]][k] + c[side[i][2]][k] + c[side[i][3]][k]) / 4.0f;
      a[k] = e[nrm[i][0]][k] - e[nrm[i][1]][k];
      b[k] = e[nrm[i][2]][k] - e[nrm[i][3]][k];
    }
    const float x[3] = {a[1] * b[2] - b[1] * a[2], a[2] * b[0] - b[2] * a[0], a[0] * b[1] - b[0] * a[1]};
    const float len = std::sqrt(dot3(x, x));
    const float n[3] = {x[0] / len, x[1] / len, x[2] / len};
    planes[i][0] = n[0];
    planes[i][1] = n[1];
    planes[i][2] = n[2];
    planes[i][3] = -dot3(n, centre);
  }
}

bool lut_is_one_to_one(const uint32_t lut_res[3], const int32_t vol_res[3])
{
  for (int a = 0; a < 3; ++a) {
    if ((int32_t)lut_res[a] != vol_res[a]) return false;
    const int n = vol_res[a];
    const float step = 1.0f / (float)n;
    for (int i = 0; i < n; ++i) {
      // the voxel-centre coordinate exactly as the integration kernel forms it
      const float s = ((float)i + 0.5f) * step;
      const float t = s * (float)n - 0.5f;
      if (t != (float)i) return false;
    }
  }
  return true;
}

void lut_z_range(int rz, int Z, int vz0, int vz1, int* lo, int* hi)
{
  const float step = 1.0f / (float)Z;
  auto texel = [&](int vz) {
    const float s = ((float)vz + 0.5f) * step;
    return (int)std::floor(s * (float)rz - 0.5f);
  };
  int a = texel(vz0) - 1, b = texel(vz1 > vz0 ? vz1 - 1 : vz0) + 2;
  if (a < 0) a = 0;
  if (b > rz - 1) b = rz - 1;
  if (a > b) a = b;
  *lo = a;
  *hi = b;
}

}  // namespace rgbdr
