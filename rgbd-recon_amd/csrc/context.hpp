// context.hpp -- private to the library: the context behind the opaque rgbdr_ctx of
// include/rgbdr.h and the helpers the translation units of the C ABI share
//   api.cpp         lifetime, frames, the per-frame calls, setters / getters, readbacks
//   api_calib.cpp   calibration volumes: upload, grid-layout arena, inverse-LUT generation
//   api_view.cpp    consumers of the volume: depth peels, ray-march, hole filling
//   api_halo.cpp    Z-slab halo staging and the RCCL exchange
//   api_timers.cpp  TimerDatabase
//   api_skip.cpp    RGBDR_FLAG_SKIP_BACKGROUND tables
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <map>
#include <string>
#include <vector>

#include "rgbdr_internal.hpp"

namespace rgbdr {
extern thread_local std::string g_create_error;

// One named interval.  In accumulate mode every begin/end takes a fresh event
// pair so a whole timed region can be resolved afterwards without a host sync
// inside it (bench.py reads the kernel's average launch duration that way).
struct Timer {
  hipEvent_t a = nullptr, b = nullptr;
  bool recorded = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending, pool;
  // intervals folded out of `pending` before rgbdr_timer_stats asked for them (a host that accumulates for hours without
  // asking must not grow an event pair per interval without bound: api_timers.cpp)
  double folded_ns = 0.0;
  uint32_t folded_count = 0;
};
}  // namespace rgbdr

namespace rgbdr {
struct PeerState;  // api_peer.cpp: the halo transport by copy engine (IPC-mapped neighbour staging sets)
void destroy_peer_state(rgbdr_ctx* ctx);
void release_peer_waits(rgbdr_ctx* ctx, int side);  // side -1: both
int peer_begin_step(rgbdr_ctx* ctx, int b, uint32_t absolute_step);  // (halo_step + 1)  // rgbdr_halo_begin_step's wait for the neighbours' copies
}

struct rgbdr_ctx {
  rgbdr_config cfg{};
  rgbdr_geometry geo{};
  int device = 0;
  hipStream_t stream = nullptr;      // where work is enqueued
  hipStream_t own_stream = nullptr;  // created with the context
  // Pipelined mode (RGBDR_FLAG_PIPELINE): upload + pre_* chain + occupied update of
  // frame k+1 run on pre_stream while integrate of frame k runs on `stream`.  The
  // only state both touch -- packed frame, occupied mask, occupied count -- is
  // double buffered; events order producer and consumer of each buffer.
  hipStream_t pre_stream = nullptr;
  std::vector<uint32_t> side_cu_mask;  // RGBDR_CU_SPLIT: the CUs of the second stream (and of the halo stream); empty: no split
  int wbuf = 0, rbuf = 0;            // buffer the next process_textures writes / the latest one written
  hipEvent_t ev_pre[2] = {nullptr, nullptr}, ev_int[2] = {nullptr, nullptr};
  // pipelined contexts: ev_view_read[b] is recorded behind the march of a view pass (rgbdr_draw) of the frame that lives in half
  // b of the frame buffers -- the last kernel that reads that frame's images, counters, mask and colour (half view_color[b] of
  // the colour frame).  The chain waits for it before it refills half b (two frames later), an upload before it rewrites that
  // colour half, the hole filling on its stream before it starts.
  hipEvent_t ev_view_read[2] = {nullptr, nullptr};
  bool ev_view_rec[2] = {false, false};
  int view_color[2] = {-1, -1};
  unsigned pre_serial = 0, pre_joined = 0;  // records of ev_pre / the last one the sweep's stream has waited for
  bool ev_pre_rec[2] = {false, false}, ev_int_rec[2] = {false, false};
  // A record on a stream costs ~20 us of its time (profiles/r06_notes/display_pipeline.md), and the one behind the sweep is
  // superseded when rgbdr_draw follows (its own record, behind the march, covers the sweep of the same frame): once a draw
  // has followed a sweep the next sweep leaves its record out (int_unrecorded), and the chain that refills that half of the
  // frame buffers records late -- behind whatever the first stream holds by then -- should no draw have come after all.
  bool draw_expected = false, int_unrecorded[2] = {false, false};
  // ev_pre / ev_int / ev_view_read are created without the system-scope fence (kernels of this device on both sides).  What
  // another device or a host library writes into the frame buffers (a sensor shard's all-gather, rgbdr_shard_view,
  // rgbdr_import_frame) needs the fence on the way to the sweep: those entry points switch the events back (system_fence_events).
  bool fenceless_events = false;
  bool pipelined() const { return (cfg.flags & RGBDR_FLAG_PIPELINE) != 0; }
  hipStream_t pstream() const { return pipelined() ? pre_stream : stream; }
  uint2* frame_buf(int b) const { return d_frame + (size_t)b * cfg.num_sensors * cfg.depth_w * cfg.depth_h; }
  uint8_t* mask_buf(int b) const { return d_mask + (size_t)b * geo.num_bricks; }
  uint32_t* count_buf(int b) const { return d_count + 4 * b; }
  std::string err;

  // images ([N][H][W][c])
  float *d_depth_raw = nullptr, *d_depth_morph = nullptr, *d_depth_rg = nullptr, *d_lab = nullptr;
  float *d_depth_b = nullptr, *d_sil = nullptr, *d_normal = nullptr, *d_quality = nullptr;
  uint2* d_frame = nullptr;
  float2* d_cc_far = nullptr;            // per pixel: frame-independent lookups of pre_depth.fs (k_pre_cache, set_calibration)
  unsigned char* d_box_flags = nullptr;
  // The colour frame has two halves (each of d_color / d_color_dxt holds both): a pipelined context uploads frame k + 1 into
  // the half the view pass of frame k does not read (everything else of a frame that two streams touch is double-buffered
  // by wbuf / rbuf / cbuf); a sequential context only ever uses half 0.
  uint8_t *d_color = nullptr, *d_color_dxt = nullptr;
  size_t color_half_bytes = 0, dxt_half_bytes = 0;
  int color_up = 0;             // the half the last upload wrote: what the next pre_* chain and the image getters read
  int color_of[2] = {0, 0};     // the half that belongs to the frame in half b of the frame buffers (the view pass's)
  bool color_consumed = false;  // a pre_* chain has read half color_up (the next upload of a pipelined context takes the other)
  uint8_t* color_half(int h) const { return d_color + (size_t)h * color_half_bytes; }
  uint8_t* dxt_half(int h) const { return d_color_dxt + (size_t)h * dxt_half_bytes; }
  bool frame_uploaded = false, textures_processed = false;
  // Sensor shard of the pre_* chain (rgbdr_set_sensor_shard): process_textures works on the layers
  // [shard_first, shard_first + shard_count) only; the packed frames of the other sensors and the other ranks' brick
  // counts arrive through rgbdr_shard_allgather (or the host's own collective on rgbdr_shard_view) before anything
  // consumes them -- shard_pending says that has not happened yet for the frame process_textures wrote last.
  int shard_first = 0, shard_count = 0;  // 0: all sensors
  bool shard_pending = false;

  // forward calibration
  float4* d_cv_xyz[rgbdr::kMaxSensors] = {};
  uint64_t inv_search_widened[rgbdr::kMaxSensors] = {};     // last inverse-LUT search of the sensor: voxels that needed a wider
  uint64_t inv_search_exhaustive[rgbdr::kMaxSensors] = {};  // window / a scan of the whole volume (rgbdr_inverse_search_stats)
  float2* d_cv_uv[rgbdr::kMaxSensors] = {};
  uint32_t xyz_res[rgbdr::kMaxSensors][3] = {}, uv_res[rgbdr::kMaxSensors][3] = {};
  float min_ds[rgbdr::kMaxSensors] = {}, max_ds[rgbdr::kMaxSensors] = {};
  float cam_pos[rgbdr::kMaxSensors][3] = {};
  float planes[rgbdr::kMaxSensors][6][4] = {};  // Frustum::getPlanes of cv_xyz
  bool have_calib[rgbdr::kMaxSensors] = {};
  bool lattice_folded[rgbdr::kMaxSensors] = {};  // cv_xyz's cells change orientation somewhere: the inverse search scans exhaustively

  // inverse calibration
  bool inv_set[rgbdr::kMaxSensors] = {};
  bool inv_tiled[rgbdr::kMaxSensors] = {};
  bool inv_resampled[rgbdr::kMaxSensors] = {};  // tiled planes hold the LUT resampled at voxel centres
  uint32_t inv_res[rgbdr::kMaxSensors][3] = {};
  float* d_lut_tiled = nullptr;       // grid-layout LUT planes of the OWNED tile layers ...
  float* d_lut_tiled_base = nullptr;  // ... inside an allocation with `halo` more layers on each side
  // that allocation as a range of mapped physical chunks (api_calib.cpp build_chunk_arena) instead of one hipMalloc
  void* lut_vmm_va = nullptr;
  size_t lut_vmm_chunk = 0;
  std::vector<hipMemGenericAllocationHandle_t> lut_vmm_handles;
  int arena_chunks = 0;               // chunks of the kept arena (0: a plain allocation)
  float arena_chunk_ms = 0.0f;        // its replay time when it was assembled
  // double_pbo of NetKinectArray (double_pixel_buffer.cpp:35-81): two page-locked host frame
  // sets; the producer fills the back one, upload_mapped swaps and DMAs from the front one
  void* h_depth[2] = {nullptr, nullptr};
  void* h_color[2] = {nullptr, nullptr};
  hipEvent_t ev_mapped[2] = {nullptr, nullptr};
  bool ev_mapped_rec[2] = {false, false};
  int mapped_back = 0;
  // Host-fed frames (rgbdr_upload_frame / rgbdr_upload_mapped_frame): the host -> device copy of frame k+1 runs on
  // copy_stream into one of two device staging sets while the passes of frame k still run; the stream that runs the
  // pre_* chain then takes the frame from the staging set with the device-resident upload (k_upload_morph: raw
  // depth + pre_morph + colour in one launch).  ev_h2d[s]: the copies into set s have landed; ev_in_read[s]: the
  // upload kernel that read set s has run (the next copy into s waits for it).
  hipStream_t copy_stream = nullptr;
  void* d_in_depth[2] = {nullptr, nullptr};
  void* d_in_color[2] = {nullptr, nullptr};
  hipEvent_t ev_h2d[2] = {nullptr, nullptr}, ev_in_read[2] = {nullptr, nullptr};
  bool ev_in_read_rec[2] = {false, false};
  int in_set = 0;                    // staging set the next host upload fills
  int32_t* d_win = nullptr;  // per (tile, sensor) frame-window origin; second plane: smallest projected depth
  uint8_t* d_skip_mask = nullptr;  // ... and per (tile, sensor) pair 1 = skipped; [ntiles * N] bytes + a 4-B counter behind them
  float* d_bgmax = nullptr;  // RGBDR_FLAG_SKIP_BACKGROUND: [N][(H+1)][(W+1)] window bounds of the current frame
  unsigned* d_skip_list = nullptr;  // {tile, verdicts, origins} of the tiles the sweep still has to work on + two counters
  unsigned* h_skip_count = nullptr; // page-locked, mapped: list length of the previous sweep
  unsigned* d_skip_count_host = nullptr;  // its device address
  int skip_parity = 0;
  bool skip_mask_valid = false;     // d_skip_mask holds the verdict bytes of the current frame
  size_t skip_mask_tiles = 0;
  float skip_limit = 0.0f;   // truncation limit the mask was built for
  int bgmax_for = -1;        // frame buffer d_bgmax was computed from since the last process_textures (-1: stale)
  float arena_probe_ms[16] = {0};  // LUT-stream time of each candidate placement of the arena
  int arena_trials = 0, arena_chosen = 0;
  float4* d_lut_generic[rgbdr::kMaxSensors] = {};
  int zoff[rgbdr::kMaxSensors] = {};

  // volume
  float *d_tsdf_base = nullptr, *d_tsdf_owned = nullptr;
  size_t layer_floats = 0;
  int halo = 0;
  float* d_linear = nullptr;  // readback scratch
  size_t linear_floats = 0;
  // The view buffers have two halves (d_view_base): d_view points at the one that holds the last ray-marched frame.  A
  // pipelined context marches frame k + 1 into the other half while the hole filling of frame k, on fill_stream, still
  // reads its own (rgbdr_draw); ev_fill[b]: behind the last side-stream fill that read half b; fill_side: such a fill is in
  // flight and whoever touches the frame or the filled image on the context's stream joins it first (join_side_fill).
  float* d_view_base = nullptr;
  int vbuf = 0;
  hipStream_t fill_stream = nullptr;
  hipEvent_t ev_fill[2] = {nullptr, nullptr};
  bool ev_fill_rec[2] = {false, false};
  bool fill_side = false;
  hipEvent_t ev_view_ready = nullptr;  // rgbdr_device_view_frame_async of a frame that is ordered on the context's stream
  float* d_view = nullptr;    // the last ray-marched frame: rgba, depth, samples (+ the first-hit indices of the slab protocol)
  size_t view_pixels = 0;
  // the depth peels have a buffer of their own (the reference draws them into m_view_depth, not into the window): a
  // stand-alone rgbdr_draw_depth_limits of any viewport leaves the frame rgbdr_fill_colors reads alone
  float* d_peels = nullptr;
  uint8_t* d_peel_near = nullptr;   // per brick: listed | near << 1, for the brick walk (kernels_raymarch.hip k_peel_near)
  size_t peel_near_cap = 0;
  size_t peel_pixels = 0;
  int view_w = 0, view_h = 0; // size of the last ray-marched frame
  float* d_fill = nullptr;    // hole filling: the LOD band of the atlas (colour + depth) and the filled frame
  size_t fill_floats = 0;
  int* d_fill_tabs = nullptr;  // tap tables of the inpaint passes for a fill_tab_w x fill_tab_h viewport (FillTabs)
  int fill_tab_w = 0, fill_tab_h = 0;
  int filled_w = 0, filled_h = 0;  // size of the frame the filled image belongs to (0: the view buffers changed since)
  rgbdr::FillTabs fill_tabs{};
  bool integrated = false;

  // bricks
  // brick counters: two buffers, switched by clearOccupiedBricks, so that a lazy occupied filter (occ_lazy) of the
  // frame before survives the next frame's clear and counting and is simply dropped when the next update replaces it
  uint32_t *d_counters = nullptr, *d_ids = nullptr, *d_count = nullptr;
  int cbuf = 0;
  uint32_t* counters_cur() const { return d_counters + (size_t)cbuf * geo.num_bricks; }
  bool clear_pending = false;       // clearOccupiedBricks was called; the zeroing rides on the next k_morph
  uint32_t* d_tile_list = nullptr;  // brick-skipping sweep: work list of owned tiles + its length (last entry)
  uint32_t* d_tile_state = nullptr; // per owned tile: epoch of the brick sweep since which it holds -limit (0: never)
  bool tile_states_kept = false;    // the last rgbdr_integrate kept them (brick / store-eliding / background-skip sweep)
  unsigned* d_empty_tiles = nullptr; // bit per tile: it and its +1 neighbours hold -limit (the ray-marcher's sample skip)
  size_t empty_tiles_cap = 0;
  int tile_count_parity = 0;        // which of the two list counters the next brick sweep appends to
  // halo staging for Z slabs: two sets of (lower face, upper face) buffers of `halo` tile layers
  float* d_stage[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
  int stage_target = -1;            // set the next integrate fills (-1: none)
  int sweep_launches = 1;           // rgbdr_set_sweep_launches
  // managed halo exchange (api_halo.cpp): side stream, per staging set "staged" / "transfer done" events
  // rgbdr_shard_allgather_async: the gather on a stream of its own, behind the chain (ev_gather_from) and in front of whoever
  // reads the completed frame (ev_gather_done: rgbdr_import_frame_from of another context, or this context's next call)
  hipStream_t gather_stream = nullptr;
  hipEvent_t ev_gather_from = nullptr, ev_gather_done = nullptr, ev_export = nullptr;
  bool gather_done_rec = false;
  // write-after-read side of rgbdr_import_frame_from: recorded on the CONSUMER's stream behind its copies out of this
  // (producer) context's frame buffer and counters; this context's next chain / import / gather waits for it
  hipEvent_t ev_imported = nullptr;
  bool imported_rec = false;
  hipStream_t halo_stream = nullptr;
  hipEvent_t ev_halo_staged[2] = {nullptr, nullptr}, ev_halo_done[2] = {nullptr, nullptr};
  bool halo_done_rec[2] = {false, false};
  bool halo_begun = false;
  bool halo_staged = false;         // an rgbdr_integrate has filled the staging set since rgbdr_halo_begin_step
  unsigned halo_step = 0;
  int halo_last = -1;
  rgbdr::PeerState* peer = nullptr;  // set by rgbdr_halo_export
  uint32_t clear_epoch = 1;         // bumped whenever the volume may have been written by anything else
  uint8_t* d_mask = nullptr;
  bool morph_current = false;       // d_depth_morph was written with the upload (k_upload_morph): the chain skips k_morph
  bool color_decoded[2] = {true, true};  // half h of d_color holds the frame uploaded into it (false: only d_color_dxt does)
  bool color_view_out = false;      // a device view of d_color was handed out: uploads keep it current
  bool mask_valid = false;
  // rgbdr_update_occupied_bricks only noted the threshold: mask_buf(rbuf) is to be rebuilt from the counters by
  // whoever needs it first -- the brick sweep's first kernel does it on the way (materialise_mask otherwise)
  bool occ_lazy = false;
  uint32_t occ_lazy_min = 0;
  int occ_lazy_cbuf = 0;            // the counter buffer the pending filter reads
  // brick -> voxel membership of divideBox / containedVoxels (geometry.cpp compute_brick_tables):
  // device copy of vox[x] | vox[y] | vox[z] | tile[x] | tile[y] | tile[z]
  rgbdr::BrickTables bt;
  uint32_t* d_brick_tab = nullptr;

  bool timers = false, accumulate = false;
  // developer A/B knob, read when the context is created
  bool separate_passes = getenv("RGBDR_SEPARATE_PASSES") != nullptr;
  bool fuse_boundary = getenv("RGBDR_NO_BOUNDARY_FUSION") == nullptr;  // A/B switch of the three-pass kernel (read once, here)
  int timer_detail = 2;  // 1: only "1preprocess" / "2integrate" / "bricks" ...; 2: also the five pre_* passes
  std::map<std::string, rgbdr::Timer> tm;

  int fail(int code, const std::string& m)
  {
    err = m;
    return code;
  }
};

// "Nothing throws" (include/rgbdr.h:15): every int-returning entry point is a function-try-block that ends in
// RGBDR_CONTAIN -- a std::bad_alloc / std::length_error from a host-side table (a corrupt LUT header, a grid of
// billions of voxels) comes back as RGBDR_ERR_NO_MEMORY, anything else as RGBDR_ERR_STATE with what() as the message.
namespace rgbdr {
int contain_exception(rgbdr_ctx* ctx) noexcept;  // only inside a catch block
}
#define RGBDR_CONTAIN(ctx) \
  catch (...) { return rgbdr::contain_exception(ctx); }

#define HIPCHK(expr)                                                                                       \
  do {                                                                                                     \
    hipError_t e_ = (expr);                                                                                \
    if (e_ != hipSuccess)                                                                                  \
      return ctx->fail(RGBDR_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));                  \
  } while (0)

#define LAUNCHCHK(what)                                                                                    \
  do {                                                                                                     \
    hipError_t e_ = hipGetLastError();                                                                     \
    if (e_ != hipSuccess) return ctx->fail(RGBDR_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e_)); \
  } while (0)

namespace rgbdr {
// device scratch that is released on every return path (HIPCHK / LAUNCHCHK return early)
struct DevScratch {
  void* p = nullptr;
  ~DevScratch() { (void)hipFree(p); }
  template <class T> T* as() const { return (T*)p; }
};

inline int nsens(const rgbdr_ctx* c) { return c->cfg.num_sensors; }
inline size_t npx(const rgbdr_ctx* c) { return (size_t)c->cfg.num_sensors * c->cfg.depth_w * c->cfg.depth_h; }

// api.cpp
int bump_clear_epoch(rgbdr_ctx* ctx);   // invalidates every recorded "this tile already holds -limit"
int sync_all(rgbdr_ctx* ctx);           // drain both streams
int join_async_gather(rgbdr_ctx* ctx, hipStream_t st);  // `st` waits for a gather rgbdr_shard_allgather_async left under way
int ensure_window_background(rgbdr_ctx* c);
int skip_sweep(rgbdr_ctx* c, IntegrateParams& p);  // the RGBDR_FLAG_SKIP_BACKGROUND sweep (p: filled by rgbdr_integrate)
int flush_clear(rgbdr_ctx* ctx);        // perform a pending clearOccupiedBricks
int materialise_mask(rgbdr_ctx* ctx);
int system_fence_events(rgbdr_ctx* ctx);  // the events between the context's streams with the default fence from now on
int wait_last_readers(rgbdr_ctx* ctx, int w, hipStream_t ps);  // before the chain's stream refills half w of the frame buffers
int ensure_color_decoded(rgbdr_ctx* ctx, int half = -1);  // RGB8 frame of a DXT upload, decoded on demand   // perform a pending (lazy) updateOccupiedBricks filter
// api_timers.cpp
void tbegin(rgbdr_ctx* c, const char* name, hipStream_t st);
void tend(rgbdr_ctx* c, const char* name, hipStream_t st);
void destroy_timers(rgbdr_ctx* c);
// api_calib.cpp
// Tile layers [t0, t1) of the grid-layout LUT that are resident: the owned layers plus
// `halo` layers on each side where the volume has them; dst = where layer t0 lives.
struct LutExtent {
  int t0, t1, vz0, vz1;
  float* dst;
};
LutExtent lut_extent(const rgbdr_ctx* ctx);
void free_lut_arena(rgbdr_ctx* c);  // api_calib.cpp: the LUT arena, a plain allocation or a range of mapped chunks
}  // namespace rgbdr
