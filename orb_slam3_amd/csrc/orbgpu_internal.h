// Internal declarations shared by the HIP translation units of liborbgpu.so.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/orbgpu.h"

#define ORBGPU_EV_SETS 32
// per-keypoint patch blur when feature_count * 43^2 < RATIO * pyramid pixels (measured: see DESIGN.md, kernel table)
#define ORBGPU_PATCH_BLUR_RATIO 0.75
#define ORB_EDGE 19        // EDGE_THRESHOLD, src/features/orb_feature_extractor.h:20
#define ORB_HALF_PATCH 15  // :18
#define ORB_PATCH 31       // :19
#define ORB_MIN_BORDER 16  // EDGE_THRESHOLD - 3, orb_feature_extractor.cpp:228

// One pyramid level as the kernels see it.  All images of a frame (pyramid, score, blurred) share
// this layout: level l starts `off` bytes into the frame slab, rows are `pitch` bytes apart.
constexpr int ORBGPU_BLUR_STRIP_CHUNKS = 7;  // k_blur: 8 * 7 - 6 = 50 output rows per strip

struct LevelGeom {
  int w, h, pitch;
  unsigned off;
  int ncols, nrows, wcell, hcell;  // FAST cell grid, orb_feature_extractor.cpp:236-242
  int cell_off;                    // first cell of this level in a frame's cell table
  int cand_off, cand_cap;          // candidate sub-pool (u32 entries) inside a frame's candidate slab
  int kp_off, kp_cap;              // keypoint slots of this level inside a frame's keypoint slab
  int xtab_off, ytab_off;          // resize coefficient tables (entries)
  int kp_size;                     // int(31 * scale_factors_[l])
  float scale_f;                   // (float)scale_factors_[l]
};

struct DeviceGeom {
  int levels;
  int fast_init, fast_min;
  unsigned frame_bytes;   // bytes of one frame slab
  int cells_per_frame;
  int cands_per_frame;
  int kps_per_frame;      // sum of kp_cap
  // compact copies of the per-level offsets (INT_MAX beyond the last level): one wide scalar load finds a level
  int kp_offs[ORBGPU_MAX_LEVELS], cell_offs[ORBGPU_MAX_LEVELS];
  LevelGeom lv[ORBGPU_MAX_LEVELS];
};

// Level 0 of the pyramid may stay in the caller's device buffer (orbgpu_extract_batch_device with rows laid out like
// the slab's): base == nullptr means it was copied into the slab like the other levels.
struct Level0Src {
  const uint8_t *base;
  long long frame_stride;
};
__device__ __forceinline__ const uint8_t *level_src(const uint8_t *pyr, const DeviceGeom &g, const Level0Src &z, int frame,
                                                    int l, unsigned level_off) {
  return (l == 0 && z.base) ? z.base + (size_t)frame * (size_t)z.frame_stride : pyr + (size_t)frame * g.frame_bytes + level_off;
}

// level of item `i` given ascending first-item offsets (offs[0] = 0): branch-free, no dependent loads
__device__ __forceinline__ int level_of(const int (&offs)[ORBGPU_MAX_LEVELS], int i) {
  int l = 0;
#pragma unroll
  for (int q = 1; q < ORBGPU_MAX_LEVELS; q++) l += (i >= offs[q]) ? 1 : 0;
  return l;
}

// resize tables: built on the host in the exact float/double sequence
struct XTab { int sx0, sx1; short a0, a1; };
struct YTab { int sy0, sy1; short b0, b1; };
// k_resize_strips: one output dword (4 pixels) of a level, as the lane that owns it needs it (48 bytes, three 16-byte loads)
struct alignas(16) RsColumn { int s0; unsigned o, keep, pad; unsigned sel[4], coef[4]; };

// packed FAST candidate: x_rel | y_rel << 12 | score << 24, coordinates relative to ORB_MIN_BORDER
__host__ __device__ inline unsigned pack_cand(int x, int y, int s) { return (unsigned)x | ((unsigned)y << 12) | ((unsigned)s << 24); }

struct CellEntry { int off, count; };  // segment of a (frame, level) candidate sub-pool

// codes kernels leave in orbgpu_ctx::d_error_flag (1, 3: the quadtree kernel's own capacity checks)
#define ORBGPU_ERR_SEGMENT_SIZE 17  // a grid range / segment is larger than the max count the call stated
#define ORBGPU_ERR_OUT_CAPACITY 16  // a frame has more keypoints than the caller's `capacity`: it was not written

// keypoints of a frame that consumers of orbgpu_extract_batch_device's output may read: a frame whose count exceeds the
// capacity was not written at all (k_orient_describe), so it counts as empty instead of exposing stale rows
__device__ __forceinline__ int usable_count(int n, int capacity) { return (n < 0 || n > capacity) ? 0 : n; }

#define HIP_TRY(expr)                                                             \
  do {                                                                            \
    hipError_t e_ = (expr);                                                       \
    if (e_ != hipSuccess) return fail_hip(ctx, e_, #expr, __FILE__, __LINE__);    \
  } while (0)

// two or three small pyramid levels built by one launch (k_resize_nest)
struct ResizeNest {
  int src_level = 0, nl = 0, ntx = 0, nty = 0;
  int pitch[3]{}, off[3]{};
  int xoff[4]{}, yoff[4]{};
  size_t lds = 0;
  short *d_xr = nullptr, *d_yr = nullptr;
};

struct orbgpu_ctx {
  orbgpu_params params{};
  DeviceGeom geom{};
  double scale[ORBGPU_MAX_LEVELS]{}, inv_scale[ORBGPU_MAX_LEVELS]{}, sigma2[ORBGPU_MAX_LEVELS]{};
  int nini[ORBGPU_MAX_LEVELS]{};
  int rs_pitch[ORBGPU_MAX_LEVELS]{}, rs_rows[ORBGPU_MAX_LEVELS]{};  // LDS footprint of a resize tile (bytes per row, rows)
  int device = 0;
  hipStream_t stream = nullptr;      // where the context enqueues: its own stream, or the caller's (orbgpu_set_stream)
  hipStream_t own_stream = nullptr;
  static constexpr int MAX_SLICES = 2;  // two slices gained 1.7 % on the headline batch, three and four lost: not kept
  hipStream_t slice_stream[MAX_SLICES]{};  // slices 1.. of a batch (slice 0 runs on `stream`)
  hipStream_t slice_l0[MAX_SLICES]{};      // level 0's FAST launch of a slice runs beside its resize chain
  hipStream_t slice_blur[MAX_SLICES]{};    // the blur of a slice runs beside its (latency-bound) quadtree kernel
  hipEvent_t slice_ev[MAX_SLICES][8]{};    // fork / blur-may-start / slice-done / level-0 fork and join / pyramid level l < early_levels done (5 ..)
  std::recursive_mutex mu;  // recursive: host-pointer entry points hold it while they call their *_device twins
  std::string err;

  // device slabs, sized for max_batch frames
  uint8_t *d_pyr = nullptr, *d_blur = nullptr;
  uint8_t *d_score = nullptr;     // score map of orbgpu_debug_dump only: allocated by the first dump that asks for it
  Level0Src lvl0{nullptr, 0};     // where the last extract call's level 0 lives (kernel argument of that call)
  mutable size_t qt_lds_configured[2] = {0, 0};  // dynamic-LDS limit already raised for k_quadtree<256> / <1024>
  bool no_zero_copy = false;      // ORBGPU_NO_ZERO_COPY=1: always copy level 0 into the slab (cross-check path)
  short *d_fast_groups = nullptr; // FastGroup table of k_fast_cells (level, cell row, first cell, cells), one frame's worth
  int n_fast_groups = 0;
  int n_fast_groups_l0 = 0;       // groups of level 0 (the table is level-major)
  int fast_group_first[ORBGPU_MAX_LEVELS + 1] = {};  // first group of every level
  int early_levels = 1;           // FAST of the levels below this on the second stream, each as soon as its level exists
  bool no_overlap = false;        // ORBGPU_NO_OVERLAP=1 (tuning aid): every kernel on the main stream, for stage times of kernels running alone
  bool early_l0 = true;           // level 0's FAST groups on the second stream beside the resize chain (ORBGPU_EARLY_L0=0: off)
  short *d_blur_strips = nullptr; // (level, first dword, first row, dwords) per strip of one frame (k_blur)
  int n_blur_items = 0;           // workgroups (8 strips) per frame
  int fast_lds_rows = 0;          // tallest cell of any level: sizes the dynamic LDS of k_fast_cells
  int n_slices = 1;               // slices a batch is cut into (ORBGPU_SLICES overrides)
  short *d_rs_strips = nullptr;   // k_resize_strips: (first dword, first row, dwords, 0) per strip, level after level
  RsColumn *d_rs_cols = nullptr;  // its column tables, level after level
  int rs_col_off[ORBGPU_MAX_LEVELS]{};
  int rs_strip_off[ORBGPU_MAX_LEVELS]{}, rs_items[ORBGPU_MAX_LEVELS]{};  // first strip / workgroups per frame (0: k_resize_level)
  ResizeNest rs_nest[4];
  int n_rs_nest = 0;
  int rs_nest_of[ORBGPU_MAX_LEVELS]{};  // nest that builds the level, or -1
  XTab *d_xtab = nullptr;
  YTab *d_ytab = nullptr;
  CellEntry *d_cells = nullptr;   // [frame][cells_per_frame]
  unsigned *d_cands = nullptr;    // [frame][cands_per_frame]
  unsigned *d_cands_packed = nullptr;  // key buffer A of the quadtree kernel when its keys exceed LDS
  unsigned *d_cands_tmp = nullptr;     // key buffer B of the device quadtree (levels too large for LDS)
  int *d_cell_prefix = nullptr;        // [frame][cells_per_frame]
  unsigned char *d_qt_nodes = nullptr; // [frame][level] node arrays of the quadtree kernel when they do not fit LDS
  unsigned *d_qt_ps = nullptr;         // [frame][qt_ps_levels] presort tables of the quadtree (k_qt_presort), big levels only
  int qt_ps_levels = 0;                // leading levels whose first quadtree rounds are presorted by several workgroups
  int qt_ps_depth[ORBGPU_MAX_LEVELS]{}; // rounds covered per level for the call being enqueued (launch_quadtree_presort)
  bool qt_ps_active = false;           // that call presorted at least one level
  bool qt_ps_blur_late = true;         // the blur forks behind the presort launches instead of behind FAST
  int *d_error_flag = nullptr;         // set by kernels on capacity violations
  int *h_error_flag = nullptr;
  short2 *d_kp_xy = nullptr;      // [frame][kps_per_frame] level coordinates after the quadtree
  int *d_kp_count = nullptr;      // [frame][levels]
  float *d_kp_angle = nullptr;    // [frame][kps_per_frame] IC_Angle in degrees (k_orient -> k_describe)
  // internal outputs used by the host-pointer entry points
  orbgpu_keypoint *d_out_kp = nullptr;
  uint8_t *d_out_desc = nullptr;
  int *d_out_n = nullptr;
  int out_capacity = 0;

  // pinned host staging
  int *h_kp_count = nullptr;
  uint8_t *h_imgs = nullptr;  // pinned upload buffer for host-pointer extraction
  uint8_t *h_out = nullptr;   // pinned download buffer of the batched host-pointer extraction: keypoints, then descriptors

  // matching scratch (grown on demand)
  void *d_match_scratch = nullptr;
  size_t match_scratch_bytes = 0;

  hipEvent_t ev[ORBGPU_EV_SETS][12]{};  // stage boundaries of the last ORBGPU_EV_SETS extract calls (ring);
                                       // [7], [8] bracket the blur on its own stream, [9] = all slices joined
  int ev_head = 0, ev_pending = 0;
  bool ev_full[ORBGPU_EV_SETS]{};      // that set holds every stage boundary, not only the FAST kernel's and the total
  int window_grid_min_from = 2048;  // orbgpu_match_window[_device]: n_from from which the search grid is used
  bool window_grid_forced = false;  // ORBGPU_WINDOW_GRID_MIN was given: the grid also where the small-problem path would run
  // host-pointer entry points stage through ONE persistent pair of buffers per context, grown on demand: pinned + mapped
  // host memory (inputs packed here, small results written here by the kernels) and its device twin (one H2D copy)
  char *h_stage = nullptr, *h_stage_dev = nullptr, *d_stage = nullptr;
  size_t stage_bytes = 0;
  // one frame per host call (orbgpu_extract): the whole call as one captured graph, results in mapped host memory
  int patch_blur_forced = -1;          // ORBGPU_PATCH_BLUR: -1 by size (patch_mode), 0 never, 1 always
  bool last_patch_mode = false;        // the last extract call blurred per keypoint (k_patch_describe): d_blur was not written
  // orbgpu_set_input_ready: the head of an orbgpu_extract_batch_device call (pyramid + FAST) on a stream of its own that
  // waits for the caller's input event, not for the context's stream.  What the head writes -- pyramid, cells,
  // candidates -- exists twice then: such calls alternate between the two WORKING SETS, so a call's head shares nothing
  // with the call before it and only waits for the last reader of its own set, two calls back (d_pyr / d_cells / d_cands
  // always name the set of the current call; the other set's buffers are allocated by the first such call and freed
  // by orbgpu_trim).
  bool inputs_ready = false;
  hipEvent_t input_event = nullptr;    // the caller's (not owned); nullptr: the images are complete when the call is made
  bool call_piped = false;             // this call qualifies (set by the entry point, read by run_pipeline)
  hipStream_t head_stream = nullptr;
  hipEvent_t ev_head_done = nullptr;   // this call's head is complete
  uint8_t *d_pyr_alt = nullptr;        // the working set that is not the current call's
  CellEntry *d_cells_alt = nullptr;
  unsigned *d_cands_alt = nullptr;
  int work_set = 0;                    // parity of the current set
  enum { SET_IDLE = 0, SET_EVENT = 1, SET_BUSY = 2 };
  // per set: nothing enqueued uses it / ev_set_free covers its last reader / in use until the context's stream drains
  int set_state[2] = {SET_IDLE, SET_IDLE};
  hipEvent_t ev_set_free[2] = {nullptr, nullptr};  // recorded on the context's stream where a call has read its set for the last time
  bool graph_single_frame = true;      // ORBGPU_NO_GRAPH=1: eager launches instead
  hipGraphExec_t graph_one = nullptr;  // upload + pipeline of one frame for graph_one_features keypoints
  int graph_one_features = 0;
  char *h_one = nullptr, *h_one_dev = nullptr;  // mapped host memory: keypoints, descriptors, count of that frame
  bool window_plain_scan = false;  // ORBGPU_WINDOW_SCAN=1: orbgpu_match_window[_device] never uses the search grid
  bool link_plain_scan = false;   // ORBGPU_LINK_SCAN=1: large-capacity links by the plain predicated scan (comparison)
  bool bruteforce_mfma = false;  // exploratory engine of the brute-force matcher (orbgpu_set_bruteforce_engine)
  bool stage_timing = false;           // record every stage boundary (each event costs a few microseconds between kernels)
  float stage_ms[8]{};
  int last_frames = 0;
  int last_feature_count = 0;
  int last_slices = 1;
};

// a vocabulary tree resident on the device of the context that created it
struct orbgpu_vocab {
  orbgpu_ctx *ctx = nullptr;
  int n_nodes = 0, levels = 0;
  int *d_child_off = nullptr;        // [n_nodes + 1]
  unsigned *d_children = nullptr;    // [child_off[n_nodes]]
  uint8_t *d_node_desc = nullptr;    // [n_nodes][32]
  unsigned *d_word_id = nullptr;     // [n_nodes]
};

int fail(orbgpu_ctx *ctx, int code, const std::string &msg);
// Tuning / test switches live in ONE environment variable, ORBGPU_TUNE="key=value;key=value" (INTEGRATION.md lists the
// keys); returns the value of `key` or nullptr.  The pointer is valid until the calling thread's next call.
const char *orbgpu_tune(const char *key);
int fail_hip(orbgpu_ctx *ctx, hipError_t e, const char *expr, const char *file, int line);

// ---- kernel launchers (each defined next to its kernels) ----
// The extraction kernels work on a slice of the batch -- frames f0 .. f0 + n - 1 of every per-frame array -- on the
// slice's stream: a batch is cut into a few slices that run side by side (run_pipeline), so that the latency-bound
// stages of one slice fill the issue slots the instruction-bound stages of another leave
struct FrameSlice { int f0, n; hipStream_t stream; };
inline Level0Src slice_lvl0(const orbgpu_ctx *c, FrameSlice fs) {
  return c->lvl0.base ? Level0Src{c->lvl0.base + (size_t)fs.f0 * (size_t)c->lvl0.frame_stride, c->lvl0.frame_stride} : c->lvl0;
}
void launch_pyramid_level(const orbgpu_ctx *c, FrameSlice fs, int level);
bool build_resize_nest(const DeviceGeom &g, const XTab *xt, const YTab *yt, int src, int nl, ResizeNest &n,
                       std::vector<short> &xr, std::vector<short> &yr);
// fused FAST score + cell NMS (k_fast.hip); score_dump != nullptr: also scatter the non-zero scores into that slab
// (laid out like the slice: its frame 0 is frame f0 of the batch)
int build_fast_groups(const DeviceGeom &g, std::vector<short> &table);
// levels [la, lb); early: the launch runs on the second stream beside the resize chain (a kernel name of its own)
void launch_fast_cells(const orbgpu_ctx *c, FrameSlice fs, uint8_t *score_dump, int la, int lb, bool early);
void launch_blur(const orbgpu_ctx *c, FrameSlice fs);
int build_blur_strips(const DeviceGeom &g, std::vector<short> &table);
int build_resize_strips(const DeviceGeom &g, int level, const XTab *xt, std::vector<short> &table, int strip_target);
void build_resize_columns(const DeviceGeom &g, int level, const XTab *xt, std::vector<RsColumn> &table);
void launch_orient(const orbgpu_ctx *c, FrameSlice fs);
void launch_describe(const orbgpu_ctx *c, FrameSlice fs, orbgpu_keypoint *d_kp, uint8_t *d_desc, int capacity,
                     int *d_n_out);
// orientation + patch blur + descriptors in one kernel behind the quadtree (no full-frame blur): large-image contexts
void launch_patch_describe(const orbgpu_ctx *c, FrameSlice fs, orbgpu_keypoint *d_kp, uint8_t *d_desc, int capacity,
                           int *d_n_out);
void launch_keypoint_geometry(const orbgpu_ctx *c, const orbgpu_camera &cam, const orbgpu_keypoint *d_kp, const int *d_n,
                              int n_frames, int capacity, double *d_undist, double *d_unproj, uint8_t *d_ok);
void launch_import_images(const orbgpu_ctx *c, const uint8_t *d_imgs, int64_t frame_stride, int stride, int n_frames);

// device quadtree (k_quadtree.hip); budget = per-level feature budget.  Returns 0 or -1 (LDS too large)
int launch_quadtree(const orbgpu_ctx *c, FrameSlice fs, const int *budget);
// the three presort launches ahead of it (no-op for contexts without presorted levels); sets qt_ps_depth / qt_ps_active
void launch_quadtree_presort(orbgpu_ctx *c, FrameSlice fs, const int *budget);
size_t quadtree_lds_bytes(int node_cap);
// dynamic LDS the quadtree kernel may ask for: the CU's 160 KB less its static arrays (scan sums, partition counters)
constexpr size_t ORBGPU_QT_LDS_LIMIT = 156 * 1024;
size_t quadtree_node_bytes(int node_cap);
size_t quadtree_presort_words();  // u32 words of presort tables per (frame, presorted level)


// matching
size_t match_scratch_size(int n_to, int n_from);
// scratch holds match_scratch_size() bytes, or -- with max_chunks limiting the split of the 'from' range -- less
void launch_match(const orbgpu_ctx *c, const orbgpu_window_args &a /* device pointers; to_xy==NULL => brute force */,
                  int *d_from_idx, unsigned *d_dist, void *scratch, int max_chunks = 0x7fffffff, bool mfma = false);
// one small search (n_to * n_from <= 4 Mi pairs): a wave per 64 x 64 block of pairs, three short launches; the outputs may
// be mapped host memory (written once, by the last launch)
bool match_small_fits(int n_to, int n_from);
size_t match_small_scratch_bytes(int n_to, int n_from);
void launch_match_small(const orbgpu_ctx *c, const orbgpu_window_args &a, int *out_from_idx, unsigned *out_dist, void *scratch);
// fixed per-GPU scratch of orbgpu_match_bruteforce_noalloc_device (a __device__ array in k_match.hip)
constexpr size_t ORBGPU_NOALLOC_SCRATCH_BYTES = 8u << 20;
constexpr int ORBGPU_NOALLOC_MAX_DEVICES = 64;
hipError_t match_noalloc_init(int device);   // with `device` current; resolves the array's address on that GPU once
void *match_noalloc_ptr(int device);
int match_noalloc_chunks(int n_to, int n_from);
size_t link_scratch_bytes(int n_pairs, int capacity);
void launch_match_frames(const orbgpu_ctx *c, const orbgpu_keypoint *d_kp, const uint8_t *d_desc, const int *d_n,
                         int capacity, const int *d_to_frame, const int *d_from_frame, int n_pairs, uint64_t window,
                         unsigned threshold, double ratio, int check_orientation, int *d_match, unsigned *d_dist,
                         int *d_n_matches);
inline size_t pad256(size_t b) { return (b + 255) & ~(size_t)255; }
size_t grid_build_scratch_bytes(int n_ranges, int max_count);
void launch_grid_build(const orbgpu_ctx *c, const double *d_xy, const int *d_ranges, int n_ranges, int max_count,
                       double image_width, double image_height, int *d_cell_start, int *d_items, void *scratch);
size_t grid_match_scratch_bytes(int n_segments, int max_to_count, int max_from_count, int n_from_total);
void launch_match_grid_segments(const orbgpu_ctx *c, const orbgpu_window_args &a, const orbgpu_segment *d_segs,
                                int n_segments, int max_to_count, int max_from_count, int n_from_total,
                                double image_width, double image_height, int *d_from_idx, unsigned *d_dist, void *scratch,
                                const orbgpu_segment *one = nullptr);
void launch_match_segments(const orbgpu_ctx *c, const orbgpu_window_args &a, const orbgpu_segment *d_segs, int n_segments,
                           int max_to_count, int n_from_total, int *d_from_idx, unsigned *d_dist, void *scratch);
size_t group_scratch_bytes(int n_from, int n_slots);
void launch_match_groups(const orbgpu_ctx *c, const orbgpu_group_args &a, int n_slots, int *d_from_idx, unsigned *d_dist,
                         void *scratch);
void launch_bow_transform(const orbgpu_ctx *c, const orbgpu_vocab *v, const uint8_t *d_desc, int n, int levelsup,
                          unsigned *d_word, unsigned *d_node);
// keyframe BowVectors resident on the device of the context that created the database: an append-only arena of
// (word, value[, count]) runs plus a per-keyframe (begin, end) table; erased runs become holes that a compaction
// pass squeezes out once they outweigh the live data
struct orbgpu_bowdb {
  orbgpu_ctx *ctx = nullptr;
  int scoring = 0;
  bool with_counts = false;
  unsigned *d_words = nullptr;
  double *d_values = nullptr;
  unsigned *d_counts = nullptr;
  long long cap = 0, used = 0, holes = 0;       // arena entries
  std::vector<unsigned long long> ids;           // slot -> keyframe id
  std::vector<long long> beg, end;               // slot -> arena run
  std::unordered_map<unsigned long long, int> slot_of;
  long long *d_beg = nullptr, *d_end = nullptr;  // device copy of the table
  size_t table_cap = 0;
  bool table_dirty = true;
  char *d_query = nullptr;                       // query words / values / counts and the two result arrays
  size_t query_cap = 0;
};

// one query against keyframe-major BowVectors: the words of keyframe k are entries kf_beg[k] .. kf_end[k])
struct BowScoreDev {
  const unsigned *q_words;
  const double *q_values;
  const unsigned *q_counts;  // may be null (= 1)
  int n_q, n_keyframes;
  const long long *kf_beg, *kf_end;
  const unsigned *kf_words;
  const double *kf_values;
  const unsigned *kf_counts;  // may be null (= 1)
  int scoring;
};
void launch_bow_score(const orbgpu_ctx *c, const BowScoreDev &a, unsigned long long *d_shared, double *d_score);
void launch_bowdb_compact(const orbgpu_ctx *c, int n_keyframes, const long long *d_old_beg, const long long *d_new_beg,
                          const long long *d_new_end, const unsigned *w_in, const double *v_in, const unsigned *c_in,
                          unsigned *w_out, double *v_out, unsigned *c_out);
void launch_distinctive(const orbgpu_ctx *c, const uint8_t *d_desc, const int *d_offsets, int n_points, int *d_best);
size_t allpairs_lds_bytes(int capacity);
// d_ang_rows / d_ang_cols: the angle (float) of keypoint 0 of keyframe 0; ang_stride: bytes between two keypoints' angles
void launch_match_allpairs(const orbgpu_ctx *c, const void *d_ang_rows, const uint8_t *d_desc_rows, const int *d_n_rows,
                           int n_rows, const void *d_ang_cols, int ang_stride, const uint8_t *d_desc_cols,
                           const int *d_n_cols, int n_cols, int capacity, int row_offset, unsigned threshold, double ratio,
                           int check_orientation, int *d_n_matches, int *d_match);
void launch_rotation_filter(const orbgpu_ctx *c, const float *d_ang_to, int n_to, const float *d_ang_from, int *d_match,
                            int *d_n_discarded);
double run_popcount_peak(const orbgpu_ctx *c);
