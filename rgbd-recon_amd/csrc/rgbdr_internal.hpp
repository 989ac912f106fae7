// rgbdr_internal.hpp -- shared between the host API (api.cpp), the host-only
// geometry code (geometry.cpp) and the kernel launchers (kernels_*.hip).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/rgbdr.h"

namespace rgbdr {

constexpr int kTile = RGBDR_TILE;          // 8
constexpr int kTileVoxels = 512;           // 8*8*8
constexpr int kMaxSensors = RGBDR_MAX_SENSORS;
constexpr int kMaxRes = 32768;           // voxels along an axis (compute_geometry)
// tsdf_raymarch.fs:34,73 marches in steps of limit / 2 for up to ceil(length / step) samples: below this a single ray takes
// millions of samples and a frame's kernel does not come back in any useful time (the reference hangs its GPU the same way)
constexpr float kMinTsdfLimit = 1.0e-6f;

// ---- host-only geometry (geometry.cpp) -------------------------------------
int compute_geometry(const rgbdr_config& cfg, rgbdr_geometry* g, std::string* err);
int slab_range(int tiles_z, int count, int rank, int* t0, int* t1);
// Brick -> voxel membership of divideBox + VolumeSampler::containedVoxels, per axis (it is
// separable): brick b holds voxel indices [first[a][b], last[a][b]]; vox[a][v] = lo | hi << 16 is
// the range of bricks holding coordinate v (lo > hi: none); overflow[a] = how far the last
// brick reaches past res - 1 (the reference then aliases other voxels through the linear index).
struct BrickTables {
  std::vector<int32_t> first[3], last[3];
  std::vector<uint32_t> vox[3];
  std::vector<uint32_t> tile[3];  // the union of vox[a] over the 8 coordinates of a storage tile, same packing
  std::vector<uint32_t> whole[3]; // 1: the tile's 8 coordinates are inside the volume and each lies in some brick
  int overflow[3] = {0, 0, 0};
};
int compute_brick_tables(const rgbdr_config& cfg, const rgbdr_geometry& g, BrickTables* t, std::string* err);
void camera_position(const float* cv_xyz, const uint32_t res[3], float out[3]);
// Frustum::getPlanes of the 8 corner samples of cv_xyz (frustum.cpp:113-177)
void frustum_planes(const float* cv_xyz, const uint32_t res[3], float planes[6][4]);
bool lattice_folds(const float* cv_xyz, const uint32_t res[3]);  // geometry.cpp
// true when an inverse LUT of resolution `lut_res` maps every voxel centre of a
// `vol_res` grid onto exactly one texel with zero interpolation weights
bool lut_is_one_to_one(const uint32_t lut_res[3], const int32_t vol_res[3]);
// z texel range [lo, hi] of a LUT with rz texels touched by voxel layers [vz0, vz1) of Z
void lut_z_range(int rz, int Z, int vz0, int vz1, int* lo, int* hi);

// ---- kernel parameter blocks -------------------------------------------------
struct PreParams {
  int N, W, H, Wc, Hc;
  // the sensor layers this launch works on: [first, first + count) -- all N unless the pre_* chain is sharded over the
  // ranks of a multi-GPU job (rgbdr_set_sensor_shard)
  int first, count;
  float bbox_min[3], bbox_max[3];
  int filter, compress;
  int refine;
  // per sensor
  const float4* cv_xyz[kMaxSensors];  // repacked to 16 B records (w = 0)
  const float2* cv_uv[kMaxSensors];
  int xyz_res[kMaxSensors][3];
  int uv_res[kMaxSensors][3];
  float cv_min_ds[kMaxSensors], cv_max_ds[kMaxSensors];
  float near_[kMaxSensors], far_[kMaxSensors];
  float cam_pos[kMaxSensors][3];
  // bricks
  float brick_size;
  int res_bricks[3];
  uint32_t* brick_counters;
  // images, all [N][H][W][channels]
  const float* depth_in;   // what the "raw_depth" unit holds for the filter pass
  const uint8_t* color;    // [N][Hc][Wc][3]
  // the same frame still in DXT1 / DXT5 blocks (null: `color` is current): pre_depth decodes its four taps itself
  const uint8_t* color_dxt;
  size_t color_layer_bytes;
  int color_mode;
  float* depth_morph;
  float* depth_rg;
  float* lab;
  float* depth_b_rg;
  float* silhouette;
  float* normal;
  float* quality;
  uint2* frame;            // packed texel for integration: {depth_b.r, quality with !silhouette in the sign bit}
  // frame-independent lookups of pre_depth.fs per pixel, filled at set_calibration (k_pre_cache)
  float2* cc_far;          // texture(cv_uv, (u, v, 1.0))
  unsigned char* box_flags;  // bit 0 / 1: texture(cv_xyz, (u, v, first / last z plane)) lies inside the box
};

struct IntegrateParams {
  int N, W, H;
  int X, Y, Z;             // full volume resolution
  int TX, TY;              // tiles per axis
  int tz0, ntz;            // first owned tile layer, number of owned tile layers
  float limit;
  float stepX, stepY, stepZ;
  const uint2* frame[kMaxSensors];
  // 1:1 mode: [local tile][sensor][3][512] floats, and per (tile, sensor) the
  // origin of the 16x16 frame window that tile projects into (int16 x | int16 y << 16)
  const float* lut_tiled;
  const int32_t* win;
  // RGBDR_FLAG_SKIP_BACKGROUND: per (tile, sensor) the smallest projected depth (k_tile_windows), per sensor and
  // window origin the largest depth of an all-background window (k_window_background); see integrate_group
  const float* win_dmin;
  const float* win_dmax;
  const int32_t* win_ext;   // 0 / 1 / 2: the footprints fit the square of 4 / 8 / 16 texels at the window origin
  const float* bgmax;
  unsigned* skip_list;       // per tile with an undecided sensor: tile, verdicts (2 bits per sensor), N window origins (k_skip_classify)
  unsigned* skip_count;      // its length (double-buffered like tile_count)
  unsigned* skip_count_next;
  volatile unsigned* skip_count_host;  // device address of a page-locked word: the list length, for the host's next grid size
  int skip_background;
  // generic mode: linear RGBA volumes, z range [zoff, zoff+nz) resident
  const float4* lut[kMaxSensors];
  int rx[kMaxSensors], ry[kMaxSensors], rz[kMaxSensors], zoff[kMaxSensors];
  // bricks
  int use_bricks;
  const uint8_t* brick_mask;
  // lazy occupied filter (k_brick_clear<true>): counters + threshold in, mask bytes out; null: the mask is current
  const uint32_t* brick_counters;
  uint32_t min_voxels;
  uint8_t* brick_mask_out;
  int num_bricks;
  int bx, by, bz;          // m_res_bricks
  // per voxel coordinate the range of bricks holding it (BrickTables::vox), on the device
  const uint32_t* vbx;
  const uint32_t* vby;
  const uint32_t* vbz;
  const uint32_t* tbx;     // the same per storage tile (BrickTables::tile)
  const uint32_t* tby;
  const uint32_t* tbz;
  const uint32_t* twx;     // BrickTables::whole per storage tile
  const uint32_t* twy;
  const uint32_t* twz;
  int ovx, ovy;            // BrickTables::overflow of x and y (indices past the z end leave the VBO: dropped)
  float* tsdf;             // first owned tile layer
  unsigned order_chunk;    // XCD-aware tile order: tiles per chunk handed to one XCD (0 = identity)
  unsigned block_base;     // full sweep issued as several launches (rgbdr_set_sweep_launches): first block of this one
  unsigned launches;       // ... and how many (0 / 1: one)
  unsigned* tile_list;     // brick-skipping sweep: owned tiles that overlap an occupied brick ...
  unsigned* tile_count;    // ... and how many (device memory, rebuilt by every sweep; zero on entry)
  unsigned* tile_count_next;  // the other of the two counters: zeroed by this sweep for the next one
  unsigned* tile_state;    // == epoch: the tile holds -limit throughout since a brick sweep of this epoch
  unsigned epoch;
  int elide_stores;        // full sweep: skip the store of an all -limit tile that tile_state says is already -limit
  // halo staging (Z slabs): the sweep also stores the tiles of its first / last `stage_layers` owned
  // tile layers into these buffers (null: that face has no neighbour), so that the transfer to the
  // neighbours need not read the volume while the next sweep overwrites it
  float* stage_lo;
  float* stage_hi;
  int stage_layers;
};

struct InvertParams {
  const float4* xyz;       // forward LUT, 16-B records, x fastest
  int rx, ry, rz;
  float planes[6][4];      // Frustum::getPlanes
  float start[3], step[3]; // sample position = start + index * step (calibration_inverter.cpp:105-125)
  int X, Y;                // inverse volume resolution in x, y
  int z0, nz;              // z rows [z0, z0 + nz) handled by this launch
  int TX, TY;              // tiles in x, y
  int window;              // index radius R of the first candidate window (widened until certified, kernels_invert.hip)
  int folded;              // the lattice folds (geometry.cpp lattice_folds): no window is certified, every voxel is scanned exhaustively
  unsigned* retry;         // (voxel, sample the walk ended at) pairs whose first window was not certified: k_invert_retry
  unsigned* retry_count;
  unsigned* todo;          // voxels (launch-relative linear index) left to k_invert_exhaustive, and their number
  unsigned* todo_count;
  unsigned long long* stats;  // [0] voxels whose window was widened, [1] voxels searched exhaustively
  float4* out_linear;      // [nz][Y][X] RGBA records, or
  float* out_tiled;        // grid layout planes of `sensor` ([tile][N][3][512], z0 must be tile aligned)
  int sensor, N;
};

struct RaymarchParams {
  // uniforms of tsdf_raymarch.fs (column-major matrices)
  float projection[16], normal_matrix[16], gl_normal_matrix_inv[16];
  float vol_to_world_inv[16], modelview_inv[16], img_to_eye[16];
  float mv_vol_to_world[16];   // gl_ModelViewMatrix * vol_to_world
  float camera_pos[3];
  int width, height, shade_mode;
  float limit;
  int N, W, H, Wc, Hc;
  int X, Y, Z, TX, TY;
  int tz_alloc0;               // global index of the first resident tile layer of tsdf / lut_tiled
  int own_z0, own_z1;          // voxel rows this context owns (slab modes)
  int res_z0, res_z1;          // voxel rows resident in this context (owned + halo)
  int* khit;                   // per pixel first-hit sample index (slab modes)
  const float* tsdf;           // tile-linear, resident layers
  const float* lut_tiled;      // grid-layout inverse LUT planes (resident layers), or null ->
  const float4* lut[kMaxSensors];
  int rx[kMaxSensors], ry[kMaxSensors], rz[kMaxSensors], zoff[kMaxSensors];
  const float2* cv_uv[kMaxSensors];
  int uv_res[kMaxSensors][3];
  const uint8_t* color;        // [N][Hc][Wc][3]
  const uint8_t* color_dxt;    // ... or the frames still in their DXT1 / DXT5 blocks (color_mode 1 / 5), color_layer_bytes each
  size_t color_layer_bytes;
  int color_mode;
  const uint2* frame[kMaxSensors];
  int skip_space;              // start positions from the depth peels (getStartPos)
  const float4* peels;
  // whole-volume march only: bit per tile, 1 = a sample whose footprint starts in the tile reads -limit from eight texels that
  // all hold it (launch_empty_tiles, from the brick sweep's tile states); null when the last sweep did not keep them or
  // the map does not fit a workgroup's LDS
  const unsigned* empty_bits;
  int empty_words;
  int whole_wave;              // the whole wavefront takes over its last long rays (march_whole_wave): 0 when that is shorter, 1 always, 2 never
  float4* out_color;
  float* out_depth;
  float* out_samples;
};
void launch_raymarch(const RaymarchParams& p, int mode, hipStream_t s);  // 0 whole volume, 1 slab find, 2 slab shade
void launch_empty_tiles(const unsigned* tile_state, unsigned epoch, int TX, int TY, int TZ, unsigned* bits, hipStream_t s);

// ReconIntegration::drawDepthLimits (glsl/bricks.{vs,gs,fs}) per pixel
struct PeelParams {
  float pmv[16];               // gl_ProjectionMatrix * gl_ModelViewMatrix
  float modelview_inv[16], img_to_eye[16];
  int width, height;
  float bbox_min[3], brick_size;
  int res_bricks[3];
  const uint32_t* counters;
  const uint8_t* mask;
  int res_super[3];            // super-cells of 4^3 bricks: ceil(res_bricks / 4)
  uint8_t* cells;              // [num_bricks] listed | near << 1, written by launch_depth_peels before the walk (k_peel_near)
  float4* out;
  // the ray-marcher's bitmap of empty tiles (launch_empty_tiles), formed by extra workgroups of k_peel_near's launch when the
  // march follows the peels: one dependent launch less in front of it (bits == null: not wanted)
  struct EmptyTiles {
    const unsigned* tile_state;
    unsigned epoch;
    int TX, TY, TZ;
    unsigned* bits;
  } empty_tiles;
};
void launch_depth_peels(const PeelParams& p, hipStream_t s);

// LOD atlas of ViewLod::setResolution (framework/rendering/view_lod.cpp:24-61)
struct FillLayout {
  int W, H, FW, num_lods;
  int off[20][2], res[20][2];
};
void make_fill_layout(int W, int H, FillLayout* L);  // geometry.cpp
// Where the 4 x 4 taps of tsdf_inpaint.fs lie, per LOD i >= 1 (the pass that writes it): for texel column fx the N column of
// the tap columns pos_int.x - 1 .. + 2 seen through framebuffer_transfer.fs's squeeze, for texel row fy the N rows.  The
// float expressions of the two shaders, evaluated on the host once per viewport size (make_fill_tables, geometry.cpp).
constexpr int FC_TAP_OUTSIDE = -1;      // texelFetch outside the texture: 0
constexpr int FC_TAP_CLEAR = -2;        // a squeezed texel beyond the LOD-0 viewport: the clear colour
constexpr int FC_ROW_CLEAR = 1 << 30;   // row flag: during pass i the band (x >= W) still holds the clear colour in this row
struct FillTabs {
  const int4* xt;  // [xbase[i] + fx] = N column of the 4 tap columns (or FC_TAP_*)
  const int4* yt;  // [ybase[i] + fy] = N row of the 4 tap rows | FC_ROW_CLEAR (or FC_TAP_OUTSIDE)
  int xbase[20], ybase[20];
  int nx, ny;      // entries in all
};
// host side of the tables: xt / yt as int quadruples, bases filled in T (T->xt / yt are the caller's to set after upload)
void make_fill_tables(const FillLayout& L, std::vector<int>* xt, std::vector<int>* yt, FillTabs* T);
// acol / adep: the atlas' column band x >= W (the LODs >= 1), fill_band_texels(L) texels each
size_t fill_band_texels(const FillLayout& L);
void launch_fill_colors(const FillLayout& L, const FillTabs& T, const float4* frame_col, const float* frame_dep, float4* acol,
                        float* adep, float4* out_col, float* out_dep, hipStream_t s);

// ---- launchers (kernels_pre.hip / kernels_integrate.hip / kernels_bricks.hip / kernels_skip.hip) ----
void launch_invert_lut(const InvertParams& p, hipStream_t s);
void launch_invert_retry(const InvertParams& p, hipStream_t s);
void launch_invert_exhaustive(const InvertParams& p, hipStream_t s);
void set_gauss_table(const float* table169);  // uploads the 13x13 spatial kernel to __constant__
void launch_u8_to_unit(const uint8_t* src, float* dst, size_t n, hipStream_t s);
void launch_decode_dxt(const uint8_t* blocks, int W, int H, int mode, int N, size_t layer_bytes, uint8_t* rgb,
                       hipStream_t s);
void launch_repack_xyz(const float* src_xyz3, float4* dst, size_t n, hipStream_t s);
void launch_morph(const PreParams& p, const float* in, float* out, uint32_t* zero, unsigned nzero, hipStream_t s);
void launch_pre_cache(const PreParams& p, int sensor, hipStream_t s);
void launch_pre_depth(const PreParams& p, uint32_t* zero, unsigned nzero, hipStream_t s);
bool launch_upload_morph(int W, int H, int first, int count, const void* depth_src, float* raw, float* morph, const void* b_src, void* b_dst,
                         size_t b_bytes, hipStream_t s);
void launch_boundary(const PreParams& p, hipStream_t s);
void launch_normal(const PreParams& p, hipStream_t s);
void launch_quality(const PreParams& p, hipStream_t s);
void launch_normal_quality(const PreParams& p, hipStream_t s);  // both passes in one launch
void launch_boundary_normal_quality(const PreParams& p, hipStream_t s);  // pre_boundary + pre_normal + pre_quality
void launch_update_occupied(const uint32_t* counters, uint32_t n, uint32_t min_voxels, uint8_t* mask,
                            uint32_t* count, hipStream_t s);
void launch_compact_occupied(const uint8_t* mask, uint32_t n, uint32_t* ids, uint32_t* count, hipStream_t s);
void launch_integrate(const IntegrateParams& p, bool one_to_one, hipStream_t s);
void launch_brick_sweep(const IntegrateParams& p, unsigned ntiles, hipStream_t s);  // kernels_bricks.hip
bool integrate_stages_halo(const IntegrateParams& p, bool one_to_one);
void launch_detile(const float* tiled, float* linear, int X, int Y, int TX, int TY, int tz0, int vz0, int vz1,
                   hipStream_t s);
void launch_tile_lut(const float4* src_rgba, int X, int Y, int Z, int src_z0, int TX, int TY, int tz0, int ntz,
                     int sensor, int N, float* dst_tiled, hipStream_t s);
void launch_untile_lut(const float* tiled, int X, int Y, int TX, int TY, int tz0, int vz0, int vz1, int sensor,
                       int N, float4* dst_rgba, hipStream_t s);
void launch_resample_lut(const float4* src_rgba, int rx, int ry, int rz, int zoff, int X, int Y, int Z, int TX, int TY,
                         int tz0, int ntz, int sensor, int N, float* dst_tiled, hipStream_t s);
// average time (ms) of one replay of the integrate kernel's LUT stream over `arena`; < 0 on error
float probe_arena_ms(const float* arena, size_t ntiles, int N, int TX, float* sink, hipStream_t s);
void launch_window_background(const uint2* frames, int W, int H, int N, float* bgmax, hipStream_t s);
void launch_skip_mask(const IntegrateParams& p, unsigned npairs, uint8_t* mask, hipStream_t s);
void launch_skip_sweep(const IntegrateParams& p, unsigned blocks, hipStream_t s);
void launch_count_bytes(const uint8_t* mask, unsigned n, unsigned* count, hipStream_t s);
void launch_tile_windows(const float* lut_tiled, int W, int H, int ntiles, int sensor, int N, int32_t* win,
                         hipStream_t s);
void launch_synth_inverse(const rgbdr_pinhole& cam, int W, int H, const float bbox_min[3], const float bbox_max[3],
                          int X, int Y, int Z, int TX, int TY, int tz0, int ntz, int sensor, int N,
                          float* dst_tiled, hipStream_t s);

}  // namespace rgbdr
