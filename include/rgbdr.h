/*
 * rgbdr.h -- C ABI of the MI355X-native TSDF-fusion + depth-preprocessing backend
 * for steppobeck/rgbd-recon.
 *
 * The reference has no FFI layer: the path sits behind the in-process C++ classes
 * kinect::NetKinectArray, kinect::CalibVolumes and kinect::ReconIntegration and
 * shares state through OpenGL texture-unit numbers.
 * This header turns every implicit binding into an explicit argument.  Each entry
 * point cites the reference interface it replaces (paths relative to the
 * reference checkout).  The library is librgbdr_hip.so; nothing here uses C++,
 * torch or HIP types.  A context is single-caller (like the GL thread that owns
 * the reference's context); all calls are asynchronous on the context's HIP
 * stream except the readbacks / getters, which synchronise it.
 *
 * Every function returns RGBDR_OK (0) or a negative rgbdr_status; nothing throws
 * across the ABI (every entry point contains C++ exceptions: std::bad_alloc comes
 * back as RGBDR_ERR_NO_MEMORY).  rgbdr_last_error() returns the message of the
 * last failure.  Grids are bounded to 32768 voxels per axis and fewer than 2^31
 * 8x8x8 tiles (rgbdr_compute_geometry refuses others), so no size computed from a
 * configuration can wrap.
 */
#ifndef RGBDR_H
#define RGBDR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The reference declares five sensor slots (sampler3D[5], glsl/tsdf_integration.vs:13; camera_positions[5],
 * camera_colors[5]): its shaders define no behaviour for a sixth sensor.  Sensors 6-8 extend the same loops by
 * their bound only and are checked against the oracle's loop bound alone -- there are no reference semantics to
 * match for N > 5 (the ray-marcher's camera-influence mode gives sensors beyond the fifth no colour). */
#define RGBDR_MAX_SENSORS 8
#define RGBDR_TILE 8        /* storage tile edge in voxels (tile-linear TSDF / LUT layout) */

typedef struct rgbdr_ctx rgbdr_ctx;

typedef enum {
  RGBDR_OK = 0,
  RGBDR_ERR_INVALID_ARGUMENT = -1, /* reference: std::invalid_argument, source/kinect_client.cpp:202,289 */
  RGBDR_ERR_OUT_OF_RANGE = -2,     /* reference: std::out_of_range, framework/NetKinectArray.cpp:248 */
  RGBDR_ERR_NO_DEVICE = -3,        /* no HIP device / HIP runtime failure at create: the product never falls back to a CPU path */
  RGBDR_ERR_HIP = -4,              /* reference: exception thrown from the GL-error callback, source/kinect_client.cpp:1051 */
  RGBDR_ERR_IO = -5,               /* reference: exit(1) on missing stream (NetKinectArray.cpp:735-738), NULL deref on missing LUT */
  RGBDR_ERR_STATE = -6,            /* call order violated (e.g. integrate before calibration was set) */
  RGBDR_ERR_NO_MEMORY = -7         /* a host-side table could not be allocated: C++ exceptions never cross this boundary (the
                                      reference lets std::bad_alloc end the process) */
} rgbdr_status;

/* flags: NetKinectArray::m_filter_textures / m_use_processed_depth / m_refine_bound
 * (framework/NetKinectArray.cpp:62-63,69, all default true) and
 * ReconIntegration::m_use_bricks (recon_integration.cpp:58, default true) */
enum {
  RGBDR_FLAG_FILTER = 1u,
  RGBDR_FLAG_PROCESSED = 2u,
  RGBDR_FLAG_REFINE = 4u,
  RGBDR_FLAG_USE_BRICKS = 8u,
  RGBDR_FLAGS_DEFAULT = 15u,
  /* not a reference setting: overlap the upload + pre_* chain of frame k+1 (second
   * HIP stream, double-buffered packed frame / occupied mask / colour frame) with integrate
   * and rgbdr_draw of frame k, and the hole filling of frame k (third stream, double-buffered
   * view buffers / filled image) with integrate and the march of frame k+1.  Results are
   * identical; only the schedule changes. */
  RGBDR_FLAG_PIPELINE = 16u,
  /* An inverse LUT whose resolution differs from the TSDF grid is normally
   * resampled once, at upload, at the voxel centres (the LINEAR lookup of
   * tsdf_integration.vs:31 is static between frames) so that integrate() streams it
   * like a 1:1 LUT.  This flag keeps the file's volume resident instead and evaluates
   * the 8-tap lookup per frame: a MEMORY-saving mode, with a measured price (profiles/r06_pmc_generic*.json; the
   * same volume bit for bit).  Four 512 x 424 sensors, full sweep:
   *   512^3 grid, 256^3 LUT (2 x coarser)     LUTs 1.16 GiB instead of 6.18; HBM 3.2 GB per sweep instead of 7.0;
   *                                           2.82 ms instead of 1.21 (VALU-bound: 1.8 G wavefront instructions against 0.53)
   *   the reference's 200 x 221 x 200 grid,   LUTs 1.70 GiB instead of 0.57 (this LUT is FINER than the grid: nothing
   *   286 x 315 x 286 LUT                     is saved); 0.58 ms instead of 0.11
   * The LUT box a tile touches is staged in LDS (k_integrate_generic_lds; 2.4 / 11 ms with plain global gathers); a LUT
   * more than ~2.5 x finer than the grid does not fit and falls back to those.  Use it where the arena is the problem
   * (1024^3 and eight sensors: 96 GiB resampled), not for speed. */
  RGBDR_FLAG_NO_RESAMPLE = 32u,
  /* Full sweep only (the brick-skipping sweep always works this way): a tile whose 512 voxels
   * all come out as -limit is not stored again when it has held -limit since an earlier sweep
   * (every voxel is still evaluated; only the store of identical bytes is left out -- 97 % of
   * the tiles of a typical scene).  Off by default so that the benchmark's full sweep performs
   * every store of the reference's clear + draw. */
  RGBDR_FLAG_ELIDE_STORES = 64u,
  /* Full sweep with a 1:1 / resampled inverse LUT only: a sensor whose frame shows nothing but background
   * (silhouette 0) in the 16x16-pixel window a tile projects into, and from which every voxel of the tile
   * lies at least the truncation limit behind the largest depth in that window, can only carve
   * (tsdf_integration.vs:34-37: tsd = -limit where tsd >= limit).  With the flag the sweep applies exactly
   * that without reading the sensor's three LUT planes of the tile (6 KiB); likewise for windows that show
   * nothing but surface in front of / behind which the whole tile lies; a tile all of whose sensors are decided
   * is a constant and is not rewritten while it holds it -- same volume bit for bit; how much is decided
   * depends on the frame (rgbdr_skipped_pairs).  Off by default so that the benchmark's
   * full sweep streams every LUT entry, like the reference's draw call samples every vertex. */
  RGBDR_FLAG_SKIP_BACKGROUND = 128u
};

/* Replaces the constructor arguments of NetKinectArray (NetKinectArray.cpp:42),
 * ReconIntegration (recon_integration.cpp:30) and the CalibrationFiles scalars
 * they read (calibration_files.cpp:26-33: element [0] decides sizes / compression
 * for all sensors). */
typedef struct {
  uint32_t struct_size;          /* sizeof(rgbdr_config), for ABI evolution */
  int32_t num_sensors;           /* CalibrationFiles::num() */
  int32_t depth_w, depth_h;      /* yml depth_size: */
  int32_t color_w, color_h;      /* yml rgb_size: */
  float bbox_min[3], bbox_max[3];/* .ks `bbx` (kinect_client.cpp:206-234), gloost::BoundingBox */
  float voxel_size;              /* ReconIntegration ctor `size` (default 0.01, kinect_client.cpp:87) */
  float brick_size;              /* setBrickSize (default 0.1) */
  float tsdf_limit;              /* ctor `limit` (default 0.01) */
  uint32_t min_voxels_per_brick; /* default 10 */
  uint32_t flags;                /* RGBDR_FLAG_* */
  int32_t compress_depth;        /* yml compress_depth: 0 = f32 metres, 1 = u8 */
  int32_t compress_rgb;          /* yml compress_rgb: 0 = RGB8, 1 = DXT1 (the yml default), 5 = DXT5 */
  float near_[RGBDR_MAX_SENSORS];/* yml near_far: per sensor (NetKinectArray.cpp:346-347) */
  float far_[RGBDR_MAX_SENSORS];
  int32_t res_override[3];       /* 0,0,0 = ceil(extent / voxel_size) as setVoxelSize does; otherwise the grid
                                    resolution per axis (voxel edge = extent / res on that axis) */
  int32_t slab_rank, slab_count; /* Z-slab of storage tiles owned by this context; 0,1 (or 0,0) = whole volume */
} rgbdr_config;

/* One calibration volume as CalibrationVolume<T> holds it
 * (framework/calibration/calibration_volume.hpp:13-84): x-fastest records. */
typedef struct {
  uint32_t res[3];
  float depth_limits[2];
  const void* data; /* host pointer: res[0]*res[1]*res[2] records of 12 B (xyz), 8 B (uv) or 16 B (xyz_inv) */
} rgbdr_lut;

/* Grid geometry derived from a config; pure host arithmetic, no device needed.
 * Restates setVoxelSize / setBrickSize / divideBox (recon_integration.cpp:341-388,
 * :474-484). */
typedef struct {
  int32_t res_volume[3];   /* m_res_volume */
  int32_t res_bricks[3];   /* m_res_bricks: the float-accumulating loops of divideBox (recon_integration.cpp:366-388) */
  float brick_size;        /* adjusted: voxel * round(size / voxel) */
  int32_t brick_voxels;    /* voxels per brick edge: round(brick_size / voxel_size) (x axis) */
  int32_t brick_voxels_axis[3]; /* per axis; differs from brick_voxels only for res_override grids with anisotropic voxels */
  int32_t num_bricks;
  int32_t tiles[3];        /* storage tiles per axis = ceil(res / RGBDR_TILE) */
  int32_t slab_tile_z0, slab_tile_z1; /* tile layers [z0, z1) owned by this slab */
  int32_t slab_voxel_z0, slab_voxel_z1; /* voxel layers [z0, z1) owned (clipped to res_volume[2]) */
  int32_t halo_tile_layers; /* 0 for a whole volume; else the tile layers of TSDF / grid-layout LUT kept on each side
                               of a slab so a ray-marcher can sample, refine and take gradients across the slab
                               faces: ceil((tsdf_limit * res_z + 2) / 8), at least 1 */
} rgbdr_geometry;

/* Which per-sensor image rgbdr_readback_image returns (the textures of
 * framework/NetKinectArray.h:76-88, units in SURVEY.md A.4). Channels in (). */
typedef enum {
  RGBDR_IMG_DEPTH_RAW = 0,   /* m_depthArray_raw as sampled: metres, or u8/255 (1) */
  RGBDR_IMG_DEPTH_MORPH = 1, /* m_textures_depth2.front (1) */
  RGBDR_IMG_DEPTH_RG = 2,    /* m_textures_depth (2) */
  RGBDR_IMG_LAB = 3,         /* m_textures_color (3) */
  RGBDR_IMG_DEPTH_B_RG = 4,  /* m_textures_depth_b (2) */
  RGBDR_IMG_SILHOUETTE = 5,  /* m_textures_silhouette (1) */
  RGBDR_IMG_NORMAL = 6,      /* m_textures_normal (3) */
  RGBDR_IMG_QUALITY = 7,     /* m_textures_quality (1) */
  RGBDR_IMG_COLOR = 8        /* m_colorArray as sampled: RGB8 (3 x u8), decoded when the frames are DXT; only through
                                rgbdr_device_image (rgbdr_readback_color returns it to the host) */
} rgbdr_image;

/* Pinhole description for the device-side synthetic inverse-LUT generator
 * (benchmark support, SURVEY.md section 8d: LUT generation for 512^3 x N must be
 * on device).  Not part of the reference surface. */
typedef struct {
  float cam_pos[3];
  float right[3], up[3], forward[3]; /* orthonormal camera axes in world space */
  float fx, fy, cx, cy;              /* pixels */
  float depth_min, depth_max;        /* metres, normalisation of the third LUT channel */
  int32_t lut_res[3];                /* forward-LUT resolution the (idx+0.5)/dims payload refers to */
} rgbdr_pinhole;

/* ---- lifetime ------------------------------------------------------------ */

/* NetKinectArray ctor + init() (NetKinectArray.cpp:42-219) and ReconIntegration
 * ctor (recon_integration.cpp:30-149): allocates every image, the TSDF volume
 * and the brick table on HIP device `device_id`. */
int rgbdr_create(const rgbdr_config* cfg, int device_id, rgbdr_ctx** out);
void rgbdr_destroy(rgbdr_ctx* ctx);
const char* rgbdr_last_error(const rgbdr_ctx* ctx); /* ctx may be NULL: last create failure */
const char* rgbdr_status_string(int status);
const char* rgbdr_version(void);

/* host-only: geometry of a config (no device touched) */
int rgbdr_compute_geometry(const rgbdr_config* cfg, rgbdr_geometry* out);
/* host-only: the voxel indices [first, last] brick `brick` of `axis` (0 x, 1 y, 2 z) holds, exactly as
 * divideBox + VolumeSampler::containedVoxels build the per-brick index lists (recon_integration.cpp:
 * 366-375, framework/rendering/volume_sampler.cpp:50-62: binary32 pos/step truncated to unsigned, float
 * upper bound) -- neighbouring bricks share a voxel row where the quotients round up, and `last` of the
 * final brick may exceed res - 1 (those indices alias other voxels through z*X*Y + y*X + x; the
 * brick-skipping sweep reproduces that).  A voxel held by no occupied brick keeps -limit. */
int rgbdr_brick_voxel_range(const rgbdr_config* cfg, int axis, int brick, int32_t* first, int32_t* last);
/* host-only: tile layers [t0,t1) of slab `rank` out of `count` over `tiles_z` layers */
int rgbdr_slab_range(int tiles_z, int count, int rank, int* t0, int* t1);
/* host-only: Frustum::getCameraPos of a cv_xyz volume (frustum.cpp:21-33, CalibVolumes.cpp:98-113) */
int rgbdr_camera_position(const rgbdr_lut* cv_xyz, float out[3]);

/* ---- calibration ---------------------------------------------------------- */

/* CalibVolumes::addVolume + createVolumeTextures (CalibVolumes.cpp:115-144):
 * uploads cv_xyz (12 B records) and cv_uv (8 B records) of one sensor; derives
 * the camera position and takes depth_limits of cv_xyz as cv_min_ds / cv_max_ds
 * (NetKinectArray.cpp:339-340). */
int rgbdr_set_calibration(rgbdr_ctx* ctx, int sensor, const rgbdr_lut* cv_xyz, const rgbdr_lut* cv_uv);
/* CalibVolumes::loadInverseCalibs (CalibVolumes.cpp:64-80): cv_xyz_inv, 16 B records. */
int rgbdr_set_inverse_calibration(rgbdr_ctx* ctx, int sensor, const rgbdr_lut* cv_xyz_inv);
/* Same, from files in the reference's on-disk format (calibration_volume.hpp:62-79);
 * any path may be NULL to skip it.  The header of a file is checked against the bytes
 * the file holds before anything is sized by it: a truncated payload, a zero or
 * absurd resolution is RGBDR_ERR_IO (the reference ignores its fread results) and
 * the context keeps the calibration it had. */
int rgbdr_load_calibration_files(rgbdr_ctx* ctx, int sensor, const char* path_cv_xyz,
                                 const char* path_cv_uv, const char* path_cv_xyz_inv);
/* CalibrationInverter::calculateInverseVolumes (framework/calibration/
 * calibration_inverter.cpp:99-155, the offline CGAL tool source/calib_inverter.cpp)
 * on the device, from the cv_xyz volume set_calibration uploaded: frustum reject ->
 * (-1,-1,-1,-1), else the inverse-distance-weighted index of the 8 nearest cv_xyz
 * samples, (index + 0.5) / dims, 1.  The k-d tree search is replaced by a local
 * search on the warped sample grid: a (2*window+1)^3 candidate window (window <= 0
 * selects the default 2) that is accepted only with a certificate that no sample
 * outside it can be among the 8 nearest, widened (up to radius 8) until it has one, and
 * replaced by a scan of the whole volume for a voxel that never gets one -- the result
 * equals the exact search for every voxel (kernels_invert.hip states the argument and
 * its assumption: a convex sampled region whose lattice does not fold).  Whether the lattice
 * folds -- its cells change orientation somewhere: a distortion model strong enough to turn the
 * image back on itself, a damaged file -- is checked when the calibration is set; such a volume
 * is searched exhaustively throughout (exact, and slow: one workgroup per voxel).
 * rgbdr_inverse_search_stats: how many voxels of the sensor's last search -- the whole call,
 * however many chunks it worked in -- needed a wider window / the exhaustive scan.
 *   compute_inverse_calibration: directly at this context's grid resolution into the
 *     resident grid layout (what `calib_inverter` + loadInverseCalibs would produce
 *     for a LUT at 1:1); with RGBDR_FLAG_NO_RESAMPLE -- every sensor of a context is resident in
 *     one layout -- as an x-fastest RGBA32F volume at the grid resolution instead, looked up per frame;
 *   generate_inverse_lut: a volume of any resolution as x-fastest RGBA32F records in
 *     host memory, ready to be written as a `.cv_xyz_inv` file. */
int rgbdr_compute_inverse_calibration(rgbdr_ctx* ctx, int sensor, int window);
int rgbdr_generate_inverse_lut(rgbdr_ctx* ctx, int sensor, const uint32_t res[3], int window, float* dst);
int rgbdr_inverse_search_stats(rgbdr_ctx* ctx, int sensor, uint64_t* widened, uint64_t* exhaustive);
/* benchmark support: fill sensor's inverse LUT on the device at 1:1 TSDF resolution */
int rgbdr_synth_inverse_calibration(rgbdr_ctx* ctx, int sensor, const rgbdr_pinhole* cam);

/* ---- per-frame calls (order is the contract, kinect_client.cpp:572-602) -- */

/* NetKinectArray::update (NetKinectArray.cpp:226-238): depth = N*H*W f32 (or u8
 * when compress_depth); color = N*Hc*Wc*3 u8, or N frames of DXT1 (8 B per 4x4
 * block) / DXT5 (16 B) blocks when compress_rgb is 1 / 5 -- decoded on the device
 * to RGB8 with the arithmetic of the reference's own CPU decoder (squish,
 * NetKinectArray.cpp:633); host pointers, copied. */
int rgbdr_upload_frame(rgbdr_ctx* ctx, const void* depth, const void* color);
/* The double pixel-buffer of NetKinectArray (double_pixel_buffer.cpp:35-81; the reader thread
 * memcpys each message into the mapped back PBO, update() swaps, NetKinectArray.cpp:226-238,
 * 511-541): rgbdr_map_frame_buffer returns the BACK one of two page-locked host frame sets for
 * the producer to fill (same layout and sizes as rgbdr_upload_frame; the sizes are returned
 * when asked); rgbdr_upload_mapped_frame swaps and enqueues the host->device copy straight
 * from those pages -- asynchronous, unlike a copy from pageable memory.  Mapping waits until
 * the copy that last read that buffer has drained. */
int rgbdr_map_frame_buffer(rgbdr_ctx* ctx, void** depth, void** color, size_t* depth_bytes, size_t* color_bytes);
int rgbdr_upload_mapped_frame(rgbdr_ctx* ctx);
/* same for buffers already resident on this context's device */
int rgbdr_upload_frame_device(rgbdr_ctx* ctx, const void* depth_dev, const void* color_dev);
/* ReconIntegration::clearOccupiedBricks (recon_integration.cpp:272-278) */
int rgbdr_clear_occupied_bricks(rgbdr_ctx* ctx);
/* NetKinectArray::processTextures (NetKinectArray.cpp:311-428): pre_morph, pre_depth (+ Lab), pre_boundary,
 * pre_normal (+ mark_brick), pre_quality.
 * Two places where the reference's shaders leave the result to the GLSL implementation; this library decides them, and a
 * maintainer comparing against the GL path will see exactly these differences:
 *  - pre_quality.fs:104-114 `pow(angle, 2.0)` with angle < 0 (a normal facing away from the sensor): undefined in GLSL
 *    (4.60 section 8.2).  Here it is the product angle * angle, i.e. a small POSITIVE quality; Mesa llvmpipe -- and any
 *    driver that lowers pow to exp2(y * log2(x)) -- returns NaN, which tsdf_integration.vs:50-55 then carries into every
 *    voxel whose 2 x 2 LINEAR footprint touches the texel.  Observed on the 4 x 512 x 424 fixtures: <= 0.5 % of the
 *    quality texels, 11 of 262 144 voxels on the 64^3 fixture; the comparisons with the Mesa run exclude those texels and
 *    voxels after checking, per texel, that the angle is negative (tests/test_gl_ref.py negative_angle).
 *  - inc_bricks.glsl:40-58 `mark_brick` of a position whose home brick lies outside the brick grid (a measurement
 *    outside the bounding box): the shader converts a negative float to uvec3 (undefined) and increments an
 *    out-of-range SSBO element (no effect under robust buffer access, memory corruption otherwise).  Here the position
 *    is SKIPPED: no counter changes.  Brick counters therefore equal the GL run's on every in-grid brick.
 * The in-grid counters themselves depend on the last bit of the world position at a brick face: at 512 x 424 they differ
 * from the Mesa run in ~3e-4 of the increments (124 of 388 134), none of them changing the occupied list. */
int rgbdr_process_textures(rgbdr_ctx* ctx);
/* ReconIntegration::updateOccupiedBricks (recon_integration.cpp:431-446), device side, no readback.  The set of
 * occupied bricks is that of the moment of the call (counters and threshold as they are now); the library may
 * evaluate the filter later, together with its first consumer. */
int rgbdr_update_occupied_bricks(rgbdr_ctx* ctx);
/* Replaces m_bricks_occupied (recon_integration.cpp:434-441) with the caller's own list of brick ids
 * (divideBox order, x fastest) -- for a host that filters the counters itself, as the reference does on
 * the CPU.  Valid until the next rgbdr_update_occupied_bricks. */
int rgbdr_set_occupied_bricks(rgbdr_ctx* ctx, const uint32_t* ids, size_t count);
/* ReconIntegration::integrate (recon_integration.cpp:243-270) */
int rgbdr_integrate(rgbdr_ctx* ctx);
/* process_textures() + integrate() of kinect_client.cpp:572-602 in one call */
int rgbdr_step(rgbdr_ctx* ctx, const void* depth, const void* color);
int rgbdr_sync(rgbdr_ctx* ctx);

/* ---- setters / getters (recon_integration.hpp:43-57, NetKinectArray.h:56-58) */

int rgbdr_set_voxel_size(rgbdr_ctx* ctx, float size);      /* reallocates volume + brick table */
int rgbdr_set_tsdf_limit(rgbdr_ctx* ctx, float limit);
int rgbdr_set_brick_size(rgbdr_ctx* ctx, float size);
int rgbdr_set_use_bricks(rgbdr_ctx* ctx, int active);
int rgbdr_set_pipelined(rgbdr_ctx* ctx, int on);           /* RGBDR_FLAG_PIPELINE at run time; drains both streams */
int rgbdr_set_elide_stores(rgbdr_ctx* ctx, int on);        /* RGBDR_FLAG_ELIDE_STORES at run time */
int rgbdr_set_skip_background(rgbdr_ctx* ctx, int on);     /* RGBDR_FLAG_SKIP_BACKGROUND at run time */
/* Issue the full sweep of rgbdr_integrate as n launches over consecutive tile ranges (default 1; the brick, listed-tile and
 * generic sweeps ignore it).  Same stores, same order within a range; between two launches the queue drains, which is when
 * a kernel waiting on ANOTHER queue -- the collective of a lagged frame gather -- gets onto the device: next to a single
 * launch that refills every wave slot as it frees, such a kernel sits until the sweep ends (profiles/r05_notes). */
int rgbdr_set_sweep_launches(rgbdr_ctx* ctx, int n);
/* (tile, sensor) pairs of the owned slab whose LUT planes a RGBDR_FLAG_SKIP_BACKGROUND sweep of the frame
 * processed last leaves unread, and the number of pairs (diagnostic: the byte accounting of such a sweep). */
int rgbdr_skipped_pairs(rgbdr_ctx* ctx, uint64_t* skipped, uint64_t* total);
/* Diagnostic copies of what RGBDR_FLAG_SKIP_BACKGROUND decides from (host buffers sized by the caller):
 *   which = 0: the verdict byte per (tile, sensor) of the frame processed last (0 read the LUT, 1 carve,
 *              2 in front of everything the window shows, 3 hidden), [owned tiles][N] bytes
 *   which = 1: per (tile, sensor) the window origin (int16 x | int16 y << 16), the smallest and the largest
 *              projected depth (-inf / +inf: a footprint leaves the window) and the size class of the square
 *              that holds the footprints (0 / 1 / 2: 4 / 8 / 16 texels), 4 planes of [owned tiles][N] words
 *   which = 2: per sensor, size class and origin the three bounds of what the square's texels have in common
 *              (largest depth if all background; smallest, largest depth if all surface), [N][3][3][H + 1][W + 1] floats */
int rgbdr_readback_skip_tables(rgbdr_ctx* ctx, int which, void* dst, size_t bytes);
int rgbdr_set_min_voxels_per_brick(rgbdr_ctx* ctx, uint32_t n);
int rgbdr_filter_textures(rgbdr_ctx* ctx, int on);         /* unlike the reference these three do not re-run */
int rgbdr_use_processed_depths(rgbdr_ctx* ctx, int on);    /* processTextures() themselves (SURVEY.md A.5:   */
int rgbdr_refine_boundary(rgbdr_ctx* ctx, int on);         /* that double-counts bricks); call it afterwards */
float rgbdr_get_brick_size(const rgbdr_ctx* ctx);
float rgbdr_occupied_ratio(rgbdr_ctx* ctx);
uint32_t rgbdr_num_bricks(const rgbdr_ctx* ctx);           /* declared but never defined in the reference (recon_integration.hpp:51) */
int rgbdr_get_geometry(const rgbdr_ctx* ctx, rgbdr_geometry* out);
int rgbdr_get_camera_position(const rgbdr_ctx* ctx, int sensor, float out[3]);

/* ---- outputs --------------------------------------------------------------- */

/* The reference has no host-side TSDF accessor (consumers sample texture unit 29).
 * Writes this context's voxel layers [slab_voxel_z0, slab_voxel_z1) as an
 * x-fastest (z*Y*X + y*X + x) f32 array (volume_sampler.cpp:57). */
int rgbdr_readback_tsdf(rgbdr_ctx* ctx, float* dst);
/* one per-sensor image, H*W*channels f32, row-major */
int rgbdr_readback_image(rgbdr_ctx* ctx, int which, int sensor, float* dst);
/* Inspection of the resident inverse LUT of one sensor as x-fastest RGBA32F records:
 * grid layout (1:1 or resampled LUT): the looked-up (u,v,d) of voxel rows [z0, z1),
 * .w = 0 (never read by the shader, tsdf_integration.vs:31, dropped at upload);
 * RGBDR_FLAG_NO_RESAMPLE: texel z rows [z0, z1) of the file's volume. */
int rgbdr_readback_inverse_calibration(rgbdr_ctx* ctx, int sensor, int z0, int z1, float* dst);
/* the colour array as sampled (RGB8, decoded when the frames are DXT): Hc*Wc*3 bytes */
int rgbdr_readback_color(rgbdr_ctx* ctx, int sensor, uint8_t* dst);
/* SSBO binding 3 payload after the 8-uint header: one u32 counter per brick */
int rgbdr_readback_brick_counters(rgbdr_ctx* ctx, uint32_t* dst);
/* m_bricks_occupied (ascending ids) and m_ratio_occupied */
int rgbdr_get_occupied(rgbdr_ctx* ctx, uint32_t* ids, size_t capacity, size_t* count, float* ratio);

/* Zero-copy consumers: device pointer of the tile-linear TSDF slab.  The owned layers are
 * read-only for the caller (the library remembers which tiles already hold -limit and does
 * not store those again); the halo layers are the caller's to fill (neighbour exchange).
 * Layout: tile (tx,ty,tz_local) at ((tz_local*tiles[1] + ty)*tiles[0] + tx)*512 floats,
 * voxel (x,y,z) inside it at (z&7)*64 + (y&7)*8 + (x&7).  `halo_layers` tile layers
 * precede and follow the owned layers when slab_count > 1. */
typedef struct {
  void* base;             /* first float of the allocation (lower halo layer if any) */
  void* owned;            /* first float of the first owned tile layer */
  size_t layer_bytes;     /* bytes of one tile layer = tiles[0]*tiles[1]*512*4 */
  int32_t owned_layers;
  int32_t halo_layers;    /* rgbdr_geometry.halo_tile_layers on each side */
} rgbdr_tsdf_device_view;
int rgbdr_device_tsdf(rgbdr_ctx* ctx, rgbdr_tsdf_device_view* out);
/* Zero-copy access to the per-sensor images for consumers that stay on the device -- what the other
 * Reconstructions of the reference sample from texture units 1-7 (ReconTrigrid:
 * framework/reconstruction/recon_trigrid.cpp:30-33 binds depth_b / quality / normals / colour; SURVEY.md
 * A.4).  `ptr` is the dense row-major image of `sensor` ([height][width][channels], f32, or u8 for
 * RGBDR_IMG_COLOR) inside the context's allocation: the same memory rgbdr_readback_image copies from,
 * valid until the context is destroyed, rewritten by every rgbdr_process_textures (colour / raw depth:
 * by every upload -- DXT frames are kept in their blocks until somebody asks for RGBDR_IMG_COLOR; from the
 * first such call on, every upload decodes them into this image on its own stream, so a view taken once
 * stays current).  The passes that write it run on `stream` (hipStream_t: the context's stream, or its
 * internal second stream under RGBDR_FLAG_PIPELINE): order a consumer after them with an event recorded
 * on that stream, or call rgbdr_sync. */
typedef struct {
  void* ptr;
  int32_t width, height, channels;
  int32_t element_bytes;  /* 4 (f32) or 1 (u8) */
  void* stream;
} rgbdr_image_device_view;
int rgbdr_device_image(rgbdr_ctx* ctx, int which, int sensor, rgbdr_image_device_view* out);
/* Zero-copy view of a sensor's calibration volumes for the drawing modes that sample them: what the reference hands
 * out as texture units -- CalibVolumes::getXYZVolumeUnits / getUVVolumeUnits, getVolumeRes, getDepthLimits
 * (framework/calibration/CalibVolumes.cpp:90-96, 146-159; consumers: recon_calibs.cpp:25-37, recon_trigrid.cpp,
 * recon_points.cpp).  cv_xyz: xyz_res[2] * xyz_res[1] * xyz_res[0] cells, x fastest, FOUR floats per cell (x, y, z and
 * an unused lane: the RGB32F records of the file repacked for 16-byte loads); cv_uv: two floats per cell.  Written by
 * rgbdr_set_calibration / rgbdr_load_calibration_files (which synchronise), constant between those calls, valid until
 * the next one for this sensor or rgbdr_destroy.  RGBDR_ERR_STATE before the sensor's calibration was set. */
typedef struct {
  const void* cv_xyz;
  const void* cv_uv;
  uint32_t xyz_res[3], uv_res[3];
  uint32_t inv_res[3];   /* resolution of the sensor's cv_xyz_inv as it was set -- sensor 0's is CalibVolumes::getVolumeRes() --,
                            0 0 0 before that; the volume itself is resident in the grid layout and is read
                            back with rgbdr_readback_inverse_calibration */
  float depth_limits[2]; /* cv_min_ds, cv_max_ds (NetKinectArray.cpp:339-340) */
  void* stream;          /* hipStream_t of the context, like the other views */
} rgbdr_calibration_device_view;
int rgbdr_device_calibration(rgbdr_ctx* ctx, int sensor, rgbdr_calibration_device_view* out);
/* Halo staging for Z slabs.  The boundary tile layers a slab sends to its neighbours are the
 * ones the next integrate overwrites, and at 1024^3 / 8 GPUs a face is 64 MiB -- a transfer as
 * long as a frame.  rgbdr_halo_staging returns two device buffers (lower face, upper face:
 * halo_tile_layers tile layers each, `bytes` long, tile-linear like the volume) of staging set
 * 0 or 1; after rgbdr_set_halo_staging(ctx, b) every rgbdr_integrate leaves a copy of its first /
 * last halo_tile_layers owned layers in set b (the full sweep stores them from the kernel
 * itself, the other sweeps copy them afterwards; a face without a neighbour is skipped), so
 * the transfer can run from there while later frames are integrated.  -1 switches it off.
 * The host alternates the two sets (rgbd-recon_amd/dist.py:HaloExchanger). */
int rgbdr_halo_staging(rgbdr_ctx* ctx, int buffer, void** lo, void** hi, size_t* bytes);
int rgbdr_set_halo_staging(rgbdr_ctx* ctx, int buffer);
/* Raw copy of resident tile layers [first, first + count) of the slab allocation (layer 0 = the lower
 * halo if any, rgbdr_tsdf_device_view.base) in the tile-linear device layout, layer_bytes each; drains
 * every stream of the context first.  For hosts without HIP that inspect the halo layers. */
int rgbdr_readback_tile_layers(rgbdr_ctx* ctx, int first, int count, float* dst);
/* The one exchange of a Z-slab step over RCCL (the reference is single-GPU, its call order
 * is source/kinect_client.cpp:572-602): grouped ncclSend / ncclRecv of the halo_tile_layers boundary
 * layers to / from the Z neighbours.  nccl_comm is the host's ncclComm_t (RCCL is bound at run time from
 * the copy loaded in the process: no link-time dependency); peer_lo / peer_hi are the communicator ranks
 * of the slabs below / above (-1: none).
 *   rgbdr_halo_exchange: one exchange on `hip_stream` (NULL: the context's stream), sending from
 *     staging set `buffer` (0 / 1, filled by the last rgbdr_integrate after rgbdr_set_halo_staging) or,
 *     with -1, straight from the owned boundary layers; receives into the halo layers of the volume.
 *   rgbdr_halo_begin_step / _exchange_async / _wait: the same with the double-buffered staging, a side
 *     stream and the events kept by the context, so that the transfer of step k overlaps step k+1:
 *     begin_step() before rgbdr_integrate, exchange_async() after it, wait() before anything on the
 *     context's stream samples across slab faces (rgbdr_raymarch_find / _shade).
 * Timer "halo" (rgbdr_timer_ns / rgbdr_timer_stats) brackets the transfer on the stream it runs on. */
int rgbdr_halo_exchange(rgbdr_ctx* ctx, void* nccl_comm, int peer_lo, int peer_hi, int buffer, void* hip_stream);
int rgbdr_halo_begin_step(rgbdr_ctx* ctx);
int rgbdr_halo_exchange_async(rgbdr_ctx* ctx, void* nccl_comm, int peer_lo, int peer_hi);
int rgbdr_halo_wait(rgbdr_ctx* ctx);
/* The same exchange by the COPY ENGINE, for the slabs of one node (SURVEY.md 8e "one exchange step per integration"; RCCL
 * stays the default transport).  A face is one contiguous range of the tile-linear layout, so a rank can PULL its
 * neighbours' staged faces with plain device-to-device copies from their staging sets, mapped once through HIP IPC: no
 * send / recv kernels next to the sweep, no collective enqueue on the per-frame path.
 *   rgbdr_halo_export(ctx, &mine)      creates the two staging sets, the interprocess events and a small POSIX shared
 *                                      memory block, and describes them in `mine` (an rgbdr_halo_peer): plain bytes, to be carried to the two
 *                                      neighbours by whatever rendezvous the host has (a file, a pipe, MPI, torch.distributed);
 *   rgbdr_halo_set_peer(ctx, side, &theirs)   maps the export of the slab below (side 0) / above (side 1); NULL: no such
 *                                      neighbour (any more).  A neighbour in the caller's own process (several contexts of
 *                                      one host, or one context standing in for its neighbours) is used directly;
 *   rgbdr_halo_begin_step / rgbdr_integrate / rgbdr_halo_pull_async / rgbdr_halo_wait   the step, as with RCCL: the copies
 *                                      run on the context's side stream behind the neighbours' "staged" events and overlap the
 *                                      next frame; every rank must call it every step (a neighbour that stops is reported as
 *                                      RGBDR_ERR_STATE after RGBDR_PEER_TIMEOUT_S seconds, default 30, not waited for forever).
 * Exporting (again) starts the protocol over: every rank exports at the same step of its loop, then sets its peers -- also
 * to come back to this transport after steps over RCCL.  A resize (rgbdr_set_voxel_size / _brick_size) releases the staging
 * sets: export and set the peers again on every rank.
 * Same node only (shared memory, HIP IPC).  Timer "halo" brackets the copies on the side stream. */
typedef struct {
  uint8_t bytes[1024];
} rgbdr_halo_peer;
int rgbdr_halo_export(rgbdr_ctx* ctx, rgbdr_halo_peer* out);
int rgbdr_halo_set_peer(rgbdr_ctx* ctx, int side, const rgbdr_halo_peer* peer);
int rgbdr_halo_pull_async(rgbdr_ctx* ctx);

/* The pre_* chain sharded by SENSOR over the ranks of a Z-slab job ("sensors are split across GPUs and
 * results all-gathered (ncclAllGather ...) with brick counters ncclAllReduce"; the reference is single-GPU and has no
 * counterpart).  After rgbdr_set_sensor_shard(ctx, first, count) rgbdr_process_textures runs the five passes for the
 * sensor layers [first, first + count) only (first = 0 with count = 0 or num_sensors: all of them again).  Uploads still take
 * every sensor's frame but bring the raw depth of the shard's layers only (colour: all); changing the shard invalidates the
 * uploaded frame (upload again before rgbdr_process_textures).  What the sweep and the slab ray-march read of a sensor is its packed frame texel (8 B per pixel),
 * and each rank has counted only its own sensors' pixels into the bricks, so before rgbdr_update_occupied_bricks /
 * rgbdr_integrate the frame is completed by
 *   rgbdr_shard_allgather(ctx, ncclComm_t)   one grouped ncclAllGather of the packed frames (in place) + ncclAllReduce(sum,
 *                                            u32) of the brick counters, enqueued on the stream the chain ran on (under
 *                                            RGBDR_FLAG_PIPELINE: next to the sweep of the frame before); rank r of the
 *                                            k ranks must hold sensors [r n / k, (r + 1) n / k); RCCL is bound at run time;
 *                                            give it a communicator of its OWN: operations on one communicator execute in
 *                                            issue order whatever their streams, so on the halo exchange's communicator the
 *                                            gather of frame k+1 would wait for the face transfer of frame k;
 *   or the host's own collectives on the device memory rgbdr_shard_view hands out (torch.distributed:
 *   rgbd-recon_amd/dist.py FrameGather), enqueued on view.stream, followed by rgbdr_shard_gather_done(ctx) -- the host's
 *   word that the frame is complete in stream order.  rgbdr_shard_view itself changes nothing: it may be called at any
 *   time to inspect the buffers.
 * Calling rgbdr_update_occupied_bricks / rgbdr_integrate on a shard before either returns RGBDR_ERR_STATE.  The float
 * images (rgbdr_readback_image / rgbdr_device_image) of the other ranks' sensors are NOT gathered: they keep what an earlier
 * unsharded frame left there.  Timer "gather" brackets the collectives. */
typedef struct {
  void* frames;           /* packed frame texels of the frame process_textures wrote last: [num_sensors][H][W] x 8 B */
  size_t sensor_bytes;    /* bytes of one sensor's layer */
  int32_t num_sensors;
  int32_t first, count;   /* the layers this context filled */
  void* counters;         /* u32[num_bricks]: this rank's brick counts (sum over ranks = the frame's) */
  uint32_t num_bricks;
  void* stream;           /* hipStream_t the chain ran on: enqueue the collectives here */
} rgbdr_shard_device_view;
int rgbdr_set_sensor_shard(rgbdr_ctx* ctx, int first, int count);
int rgbdr_shard_view(rgbdr_ctx* ctx, rgbdr_shard_device_view* out);
int rgbdr_shard_allgather(rgbdr_ctx* ctx, void* nccl_comm);
int rgbdr_shard_gather_done(rgbdr_ctx* ctx);
/* The frame a context sweeps may come from ANOTHER context's pre_* chain: packed frame texels of every sensor
 * ([num_sensors][H][W] x 8 B) and, optionally, the u32 brick counters ([num_bricks]) in device memory -- what
 * rgbdr_shard_view of the producing context hands out after its gather.  Enqueued on the chain's stream behind
 * wait_event (a hipEvent_t, or NULL); afterwards the context is where rgbdr_process_textures would have left it
 * (rgbdr_update_occupied_bricks / rgbdr_integrate next).  The float images of rgbdr_readback_image are NOT part of
 * it.  Producer and consumer must agree in sensors, image size, bounding box and brick size (the brick grid); the
 * producer's voxel size is free.  Use: a chain-only context runs frame k+1 while this one sweeps frame k, so the
 * all-gather of frame k+1 travels under that sweep (rgbd-recon_amd/dist.py LaggedChain; the reference has no
 * counterpart: it is single-GPU, kinect_client.cpp:572-602).
 * Ordering: the copies are enqueued on the stream THIS context's chain runs on (its second stream under
 * RGBDR_FLAG_PIPELINE).  With the pointer form the caller keeps the source unchanged until they have run -- producer and
 * consumer on one stream, or an event recorded behind this call that the source's next writer waits for;
 * rgbdr_import_frame_from does both sides itself: the copies wait for the producer's chain and gather, and the producer's
 * next rgbdr_process_textures / rgbdr_import_frame* / rgbdr_shard_allgather waits for the copies, whatever the streams. */
int rgbdr_import_frame(rgbdr_ctx* ctx, const void* packed_frames, const void* brick_counters, void* wait_event);
/* The same without a HIP type at the boundary (what a C / C++ host uses, host::LaggedChain): the gather of the producing
 * context on a stream of that context's own -- behind its chain, with an event behind it that every later call touching
 * the frame (on either context) waits for --, and the import straight from the producer, in the order of the producer's
 * chain and of that gather.  Both contexts live on one device and agree in sensors, image size and brick grid. */
int rgbdr_shard_allgather_async(rgbdr_ctx* ctx, void* nccl_comm);
int rgbdr_import_frame_from(rgbdr_ctx* ctx, rgbdr_ctx* producer);
/* device pointer of the packed per-sensor frame the integration kernel samples:
 * H*W 8-byte texels {f32 depth_b.r, f32 quality with (silhouette == 0) in the sign bit} */
int rgbdr_device_frame(rgbdr_ctx* ctx, int sensor, void** ptr);
/* the context's HIP stream (hipStream_t) for callers that enqueue dependent work */
void* rgbdr_stream(rgbdr_ctx* ctx);
/* Enqueue all further work of this context on the caller's stream (hipStream_t of
 * the same device; NULL restores the context's own stream).  Lets a host that owns
 * other work on that stream -- e.g. RCCL halo exchanges -- order it against the
 * kernels without host synchronisation.  The previous stream is drained first. */
int rgbdr_set_stream(rgbdr_ctx* ctx, void* hip_stream);

/* ---- ray-marching the volume (consumer of the TSDF) ---------- */

/* The uniforms ReconIntegration::draw uploads for glsl/tsdf_raymarch.{vs,fs}
 * (framework/reconstruction/recon_integration.cpp:177-241), computed by the host
 * exactly as there (glm / gloost); all matrices column-major as glGetFloatv returns
 * them.  Every matrix entry and the camera position must be finite: a NaN or an infinity
 * is RGBDR_ERR_INVALID_ARGUMENT (tsdf_raymarch.fs:86 would turn it into four billion
 * samples per ray). */
typedef struct {
  float modelview[16];           /* GL_MODELVIEW_MATRIX */
  float projection[16];          /* GL_PROJECTION_MATRIX */
  float normal_matrix[16];       /* inverseTranspose(modelview * vol_to_world), :200-201 */
  float gl_normal_matrix_inv[16];/* inverse(gl_NormalMatrix), only read in shade mode 2 (shading.glsl:66) */
  float vol_to_world[16];        /* translate(bbox_min) * scale(bbox extent), recon_integration.cpp:66-72 */
  float vol_to_world_inv[16];    /* the shader calls inverse() on these two per fragment (:387-388) */
  float modelview_inv[16];
  float img_to_eye[16];          /* inverse(viewport_scale * viewport_translate * projection), :185-194 */
  float camera_pos[3];           /* camera in volume space, :203-206 */
  int32_t width, height;         /* viewport */
  int32_t shade_mode;            /* UBO Settings.g_shade_mode: 0 colour, 1 shaded, 2 normal, 3 camera influence */
  int32_t skip_space;            /* m_skip_space && m_use_bricks: start each ray at the brick depth peels */
} rgbdr_view;

/* ReconIntegration::draw: one ray per pixel through the whole TSDF volume (step
 * limit/2, secant refinement at the zero crossing, gradient normal, quality /
 * (distance + 0.01) colour blend).  Outputs (host pointers, any may be NULL):
 * color = height*width RGBA32F, depth = gl_FragDepth, num_samples = the
 * tex_num_samples image; pixels the ray-marcher discards keep the cleared values of
 * ViewLod::enable: (0,1,0,0), depth 1.  Needs the whole volume in this context
 * (slab_count == 1; for Z slabs see rgbdr_raymarch_find / _shade) and a completed
 * integrate(). */
int rgbdr_raymarch(rgbdr_ctx* ctx, const rgbdr_view* view, float* color, float* depth, float* num_samples);

/* Ray-marching a volume that is split into Z slabs (one context per GPU).  Every rank
 * marches the same sample sequence per pixel; a sample belongs to the slab that owns the
 * voxel row its LINEAR footprint starts at.  rgbdr_raymarch_find writes, per pixel, the
 * index of the first sample this slab owns with density > 0 (0x7fffffff: none) into a
 * device buffer of height*width int32; the host takes the element-wise MINIMUM of that
 * buffer over all slabs (e.g. all-reduce MIN over RCCL, in place); rgbdr_raymarch_shade
 * then refines and shades exactly the pixels whose minimum this slab owns, returns the
 * cleared values elsewhere, and overwrites the buffer with 0x7fffffff where it did not
 * shade -- so the frames of all slabs composite by selection, bit-identical to
 * rgbdr_raymarch on a single context.  Needs the TSDF halo layers to be current
 * (rgbdr_device_tsdf + the neighbour exchange of rgbd-recon_amd/dist.py). */
int rgbdr_raymarch_find(rgbdr_ctx* ctx, const rgbdr_view* view, void** first_hit_device);
int rgbdr_raymarch_shade(rgbdr_ctx* ctx, const rgbdr_view* view, float* color, float* depth, float* num_samples);

/* ReconIntegration::drawDepthLimits (recon_integration.cpp:409-429, glsl/bricks.*): the
 * occupied bricks' depth peels for `view`: height*width RGBA32F texels (nearest face z,
 * -farthest face z, nearest back-face z, 0), cleared value (1,0,1,0).  rgbdr_raymarch runs
 * it by itself when view->skip_space is set; this entry point exposes the image.  The peels
 * live in a buffer of their own (the reference's m_view_depth): drawing them, for any
 * viewport, leaves the last ray-marched frame -- what rgbdr_fill_colors fills -- untouched. */
int rgbdr_draw_depth_limits(rgbdr_ctx* ctx, const rgbdr_view* view, float* peels);

/* ReconIntegration::fillColors (recon_integration.cpp:280-339): screen-space hole
 * filling of the frame the last rgbdr_raymarch produced -- the tsdf_inpaint.fs pyramid
 * over the ViewLod atlas, then tsdf_colorfill.fs (first LOD with alpha > 0, blended
 * with the next two).  color = height*width RGBA32F, depth = height*width.
 * These are the fragment shader's OUTPUTS for every pixel.  The reference draws that pass into its window with
 * GL_LESS against the cleared depth (recon_integration.cpp:314, kinect_client.cpp:614,994): a fragment whose depth is
 * exactly 1 -- a ray that hit nothing -- fails the test and the window keeps its clear colour there (seen in the run
 * of the shaders on Mesa, tests/test_gl_ref.py).  A consumer that wants the reference's window takes
 * depth < 1 ? color : its clear colour. */
int rgbdr_fill_colors(rgbdr_ctx* ctx, float* color, float* depth);
/* ReconIntegration::drawF (recon_integration.cpp:151-178), the displayed frame of the reference's main loop
 * (kinect_client.cpp:572-617: update, process_textures, integrate, drawF): drawDepthLimits when view->skip_space
 * ("brickdraw"), Reconstruction::drawF = the ray-march ("draw"), fillColors when fill_holes ("holefill"; m_fill_holes),
 * the whole under "3recon".  Enqueued on the context's stream behind the frame's passes WITHOUT waiting for the device
 * (contexts with a halo or gather transfer on a side stream are drained first); the frame stays on the device like the
 * window's framebuffer does.  With RGBDR_FLAG_PIPELINE the pass is ordered against the chain's stream by events: the
 * pre_* chain of the NEXT frame, its upload included, runs under it (the pass reads the halves of the double buffers --
 * images, brick counters and masks, colour frame -- that chain does not write), and the hole filling runs on a stream of
 * its own, the next frame's sweep, peels and march under it (the view buffers have two halves as well).  The filled frame
 * is then ordered on that stream: rgbdr_device_view_frame, rgbdr_readback_view_frame and rgbdr_fill_colors make the
 * context's stream wait for it, so what they return is ordered on rgbdr_set_stream's stream as ever -- a host that asks
 * for every frame this way gets the frames one after the other (that wait sits in front of the next frame's sweep); the
 * overlap is for a host that lets frames go by -- a throughput run, a recorder that keeps every n-th frame.  rgbdr_device_view_frame returns where: colour [height][width][4] f32 and
 * depth [height][width] f32 of the ray-marched frame (filled = 0) or of the hole-filled one (filled = 1; RGBDR_ERR_STATE
 * when the current frame has not been filled), valid until the next call that draws, fills or uploads a frame and ordered
 * on rgbdr_set_stream's stream.  rgbdr_readback_view_frame copies it to the host (waits). */
int rgbdr_draw(rgbdr_ctx* ctx, const rgbdr_view* view, int fill_holes);
int rgbdr_device_view_frame(rgbdr_ctx* ctx, int filled, void** color, void** depth, int* width, int* height);
/* The same for a host that presents from a queue of its own: nothing is made to wait; *ready_event (a hipEvent_t owned by
 * the context) completes when the frame is written -- behind the hole filling on its stream, or behind what the
 * context's stream holds -- and the host's queue waits for it (hipStreamWaitEvent) before it reads.  The filled image
 * and the view buffers have two halves: the pointers of frame k stay untouched until the rgbdr_draw of frame k + 2 is
 * called, so a presenter that is done with frame k by then overlaps with the library frame after frame. */
int rgbdr_device_view_frame_async(rgbdr_ctx* ctx, int filled, void** color, void** depth, int* width, int* height, void** ready_event);
int rgbdr_readback_view_frame(rgbdr_ctx* ctx, int filled, float* color, float* depth);

/* Placement of the inverse-LUT arena.  The integrate sweep time depends on where the
 * driver placed that allocation (stable per allocation, up to 12 % apart: a zone of 13-19 GB of the device memory,
 * usually where the first large allocation of a process lands, streams slower than the rest), so the
 * library times the LUT stream on up to RGBDR_ARENA_TRIALS candidate allocations when the arena is created
 * (environment; 1..16; default 16 for arenas of 1 GiB and more, 1 = the first allocation is taken and nothing is
 * probed for smaller ones) -- held while probing, i.e. up to n x the arena of HBM transiently and never more than
 * what leaves 4 GiB free, at most about 1 s -- stops at the first that streams at the fast level and otherwise
 * keeps the fastest.  bench.py runs on this default and reports every candidate.  Reports
 * the candidates' times in ms (0 where none was measured), how many were tried and which
 * one was kept. */
int rgbdr_get_arena_probe(const rgbdr_ctx* ctx, float ms[16], int* trials, int* chosen);
/* Where no candidate allocation reached the fast level, the library assembles the arena from the fastest of a pool
 * of 1-GiB physical chunks (HIP's virtual-memory API: every chunk timed with the sweep's streams, the fastest mapped into
 * one range, the rest released; physical memory streams at two levels per chunk, profiles/r05_vmm_chunk_map.txt) and
 * keeps that range if its replay beats the best plain candidate's (~25 ms of set-up per chunk of the pool, which is twice
 * the arena or what leaves 4 GiB free).  Environment: RGBDR_ARENA_CHUNKS=0 never,
 * =force always; RGBDR_ARENA_CHUNK_MB the chunk size.  Reports the chunks of the kept arena (0: a plain allocation) and
 * its replay time in ms when it was assembled. */
int rgbdr_get_arena_chunks(const rgbdr_ctx* ctx, int* chunks, float* ms);
/* Device memory released shortly before (by this or by an earlier process) is wiped by the
 * driver in the background and slows every stream for a moment.  rgbdr_settle replays the
 * integrate kernel's LUT-read + TSDF-store stream (the volume's contents are undefined
 * afterwards, until the next integrate) until it runs at the fastest level of the hardware
 * (>= 6.55 TB/s) or max_seconds have passed (at most 60; a negative or non-numeric budget is
 * RGBDR_ERR_INVALID_ARGUMENT), and returns the last replay's time in ms.  For benchmarks and
 * latency-critical start-up; never needed for correctness. */
int rgbdr_settle(rgbdr_ctx* ctx, float max_seconds, float* stream_ms);

/* Makes a frame composited from several slab contexts (rgbdr_raymarch_shade + selection)
 * the "last ray-marched frame" of this context, so that rgbdr_fill_colors can run on it
 * (the framebuffer ReconIntegration::fillColors reads, recon_integration.cpp:283-296).
 * color = height*width RGBA32F, depth = height*width, host memory. */
int rgbdr_upload_view_frame(rgbdr_ctx* ctx, int width, int height, const float* color, const float* depth);

/* ---- timers (TimerDatabase, framework/rendering/timer_database.cpp:26-49) -- */

/* names: "morph","bilateral","boundary","normal","quality","1preprocess","2integrate","bricks",
 * "draw","brickdraw","holefill" (timer_database.cpp:26-49, reconstruction.cpp:35-39,
 * recon_integration.cpp:161-163,410-428) and "halo" (the slab exchange).  Last completed interval, ns. */
int rgbdr_enable_timers(rgbdr_ctx* ctx, int on);
int rgbdr_timer_ns(rgbdr_ctx* ctx, const char* name, uint64_t* ns);
/* Accumulating mode: every interval of a timed region keeps its own HIP event
 * pair (recorded on the context's stream, no host sync inside the region);
 * rgbdr_timer_stats synchronises once, returns the summed duration and the
 * number of intervals since the previous call, and resets them
 * (TimerDatabase running mean, timer_database.cpp:59-121). */
int rgbdr_enable_timer_accumulation(rgbdr_ctx* ctx, int on);
/* 2 (default): all timers; 1: only the totals ("1preprocess", "2integrate", "bricks",
 * "draw", ...), not the five pre_* passes inside "1preprocess"; 0: "2integrate" alone --
 * every timer costs two event records (~4 us of stream time each) between small kernels. */
int rgbdr_set_timer_detail(rgbdr_ctx* ctx, int detail);
int rgbdr_timer_stats(rgbdr_ctx* ctx, const char* name, uint64_t* total_ns, uint32_t* count);

#ifdef __cplusplus
}
#endif
#endif /* RGBDR_H */
