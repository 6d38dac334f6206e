/*
 * orbgpu.h -- C ABI of the MI355X-native ORB front-end (liborbgpu.so).
 *
 * This is the drop-in boundary for the ORB extraction + descriptor-matching path of
 * vahagnIV/orb_slam3.  The reference has no FFI layer; its seam is the C++ abstraction
 * features::IFeatureExtractor (src/features/ifeature_extractor.h:24-60) plus the header-only
 * matcher features::matching::SNNMatcher<It> (src/features/matching/second_nearest_neighbor_matcher.hpp:19-84)
 * and features::matching::OrientationValidator (src/features/matching/validators/orientation_validator.h:19-32).
 * The C++ adapters in orb_slam3_amd/host/ keep those names and signatures and call the entry points
 * below; INTEGRATION.md shows the binding a maintainer adds on the reference side.
 *
 * Conventions: plain pointers and sizes, caller-allocated outputs, int status (0 = ok, < 0 = error,
 * message via orbgpu_last_error), no exceptions cross the boundary.  Entry points that end in
 * `_device` take DEVICE pointers (hipMalloc memory on the context's GPU) and run on the context's
 * stream without synchronising the host unless stated; all others take HOST pointers.
 * A context serialises its own calls (internal mutex); use one context per calling thread to overlap.
 * There is no CPU fallback: every entry point fails with ORBGPU_E_HIP when no GPU is usable.
 */
#ifndef ORBGPU_H_
#define ORBGPU_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORBGPU_MAX_LEVELS 16

enum {
  ORBGPU_OK = 0,
  ORBGPU_E_EMPTY = -1,    /* empty image: the value Extract returns (orb_feature_extractor.cpp:533) */
  ORBGPU_E_INVALID = -2,  /* bad argument / unsupported geometry */
  ORBGPU_E_HIP = -3,      /* HIP runtime error or no device */
  ORBGPU_E_CAPACITY = -4, /* caller buffer or context capacity too small */
};

typedef struct orbgpu_ctx orbgpu_ctx;

/* Constructor arguments of ORBFeatureExtractor (src/features/orb_feature_extractor.h:24-29) plus
 * device placement and batch sizing. */
typedef struct {
  int32_t width, height;
  double scale_factor;
  int32_t levels;
  int32_t fast_init, fast_min; /* init_threshold_FAST / min_threshold_FAST */
  int32_t device;              /* HIP device ordinal */
  int32_t max_batch;           /* frames per orbgpu_extract_batch* call (>= 1) */
  int32_t max_features;        /* largest feature_count that will be requested per frame.  When the share of one
                                * level (orb_feature_extractor.cpp:211-224) exceeds about 2060 keypoints -- max_features
                                * above about 9400 at 8 levels x 1.2, above 2060 with a single level -- the quadtree
                                * kernel keeps its node lists in global scratch instead of LDS (same results, slower) */
} orbgpu_params;

/* features::KeyPoint (src/features/key_point.h:10-23) in wire form.  angle holds the float that
 * cv::fastAtan2 returned (the reference widens it to double); size = int(31 * scale_factors_[level]). */
typedef struct {
  double x, y;
  float angle;
  int32_t size;
  int32_t level;
  int32_t reserved;
} orbgpu_keypoint;

/* Derived geometry (orb_feature_extractor.cpp:53-69, 211-242, 506-508); budget is for the
 * feature_count passed to orbgpu_get_geometry. */
typedef struct {
  int32_t levels;
  int32_t w[ORBGPU_MAX_LEVELS], h[ORBGPU_MAX_LEVELS], budget[ORBGPU_MAX_LEVELS];
  int32_t ncols[ORBGPU_MAX_LEVELS], nrows[ORBGPU_MAX_LEVELS], wcell[ORBGPU_MAX_LEVELS], hcell[ORBGPU_MAX_LEVELS];
  int32_t kp_size[ORBGPU_MAX_LEVELS];
  double scale[ORBGPU_MAX_LEVELS], inv_scale[ORBGPU_MAX_LEVELS], sigma2[ORBGPU_MAX_LEVELS];
  int64_t pyramid_pixels;     /* P = sum of level pixels */
  int32_t keypoint_capacity;  /* upper bound of keypoints per frame for this feature_count */
} orbgpu_geometry;

/* ---- lifetime -------------------------------------------------------------------------------- */
/* replaces: new ORBFeatureExtractor(w, h, scale, levels, initFAST, minFAST)  (main.cpp:451-454,
 * factories/feature_extractor_factory.cpp:10-17) */
int orbgpu_create(const orbgpu_params *params, orbgpu_ctx **out_ctx);
/* replaces: delete feature_extractor_ (map/atlas.cpp:92-94) */
void orbgpu_destroy(orbgpu_ctx *ctx);
/* message of the last failing call on this context (or of a failed orbgpu_create when ctx == NULL) */
const char *orbgpu_last_error(const orbgpu_ctx *ctx);
int orbgpu_get_geometry(const orbgpu_ctx *ctx, int32_t feature_count, orbgpu_geometry *out);
/* The context enqueues all its work on one stream (a hipStream_t, returned here): a stream of its own unless
 * orbgpu_set_stream gave it the caller's.  Device buffers that the caller fills or reads on another stream (torch's,
 * say) must be ordered against it: orbgpu_set_stream / orbgpu_wait_event / orbgpu_record_event below, or a
 * synchronisation of that stream before a *_device call and orbgpu_synchronize before reading the results. */
void *orbgpu_stream(orbgpu_ctx *ctx);
int orbgpu_synchronize(orbgpu_ctx *ctx);
/* no counterpart in the reference (a scheduling hint for callers that feed batch after batch; results and their order
 * on the context's stream are unchanged).
 * By default the kernels of an extract call start behind everything already enqueued on the context's stream -- which is
 * also how they see the caller's images.  With ready != 0 the caller promises that the images of every later
 * orbgpu_extract_batch_device call are complete when the call is made, or, with a non-null `event` (a hipEvent_t), when
 * that event has fired (call this before each extract to name the batch's own event).  The library then keeps what a
 * call's first kernels write (pyramid, FAST cells, candidates) twice, lets such calls alternate between the two working
 * sets, and runs a call's pyramid and FAST kernels on a stream of its own -- beside the previous batch's quadtree,
 * orientation and descriptor kernels and whatever the caller enqueued behind them (the link) -- while everything from
 * the quadtree on, and so the outputs, stays in the context's stream order.  The second working set (as large as the
 * pyramid slab) is allocated by the first such call and given back by orbgpu_trim.  Applies to images that are read in
 * place (orbgpu_extract_batch_device's layout rule) with the full-frame blur; other calls run as before.
 * ready == 0 (the default) restores the plain stream order. */
int orbgpu_set_input_ready(orbgpu_ctx *ctx, int32_t ready, void *event);

/* No counterpart in the reference.  Waits for the context's work (like orbgpu_synchronize, whose status it returns) and
 * releases the side streams an extract call forks onto (level 0's FAST launch, the blur, batch slices) together with
 * their events; the next extract call re-creates the ones it needs.  A process maps all its streams onto 4 hardware
 * queues: a caller that keeps several contexts alive but extracts with one at a time trims the idle ones. */
int orbgpu_trim(orbgpu_ctx *ctx);
/* Ordering against the caller's streams without a device-wide synchronisation (no counterpart in the reference, which
 * never leaves the CPU).  Either
 *  - orbgpu_set_stream: the context enqueues on the caller's stream (a hipStream_t; NULL = back to its own stream)
 *    from now on, so the caller's fills, the library's kernels and the caller's reads are ordered by the stream itself
 *    (work already enqueued on the previous stream is waited for first); or
 *  - orbgpu_wait_event: the context's stream waits for a hipEvent_t the caller recorded after filling its buffers, and
 *    orbgpu_record_event: records the caller's hipEvent_t on the context's stream, for the caller's stream to wait on
 *    before it reads the results.
 * The stream / event must belong to the context's device and outlive the work that uses it. */
int orbgpu_set_stream(orbgpu_ctx *ctx, void *stream);
/* A batch can be extracted as two slices of frames that run side by side on streams forked from (and joined back into)
 * the context's stream: the latency-bound tail of one slice (quadtree, orientation) then overlaps the FAST kernel of the
 * other (+1.7 % on 256 VGA frames).  n_slices = 0 or 1 (the default): one slice -- what a per-kernel profile wants, since
 * the stage timers then see one kernel at a time; 2: two slices. */
int orbgpu_set_batch_slices(orbgpu_ctx *ctx, int32_t n_slices);
int orbgpu_wait_event(orbgpu_ctx *ctx, void *event);
int orbgpu_record_event(orbgpu_ctx *ctx, void *event);

/* ---- extraction ------------------------------------------------------------------------------ */
/* replaces: IFeatureExtractor::Extract (ifeature_extractor.h:33-35; ORBFeatureExtractor::Extract
 * orb_feature_extractor.cpp:530-600) for one image.  Keypoints and descriptors come out in the
 * reference's order (reversed packing, :550,579-587).  *n_out = number of keypoints.
 * Returns ORBGPU_E_EMPTY for an empty image like the reference's -1. */
int orbgpu_extract(orbgpu_ctx *ctx, const uint8_t *img, int32_t width, int32_t height, int32_t stride,
                   int32_t feature_count, orbgpu_keypoint *kp, uint8_t *desc, int32_t capacity, int32_t *n_out);
/* Same for n_frames images (<= max_batch) in one pass over the GPU; frame f writes kp[f*capacity ...],
 * desc[f*capacity*32 ...], n_out[f].  The reference calls Extract once per frame
 * (factories/feature_handler_factory.cpp:50); batching is what the offline workloads use. */
int orbgpu_extract_batch(orbgpu_ctx *ctx, const uint8_t *const *imgs, int32_t n_frames, int32_t stride,
                         int32_t feature_count, orbgpu_keypoint *kp, uint8_t *desc, int32_t capacity,
                         int32_t *n_out);
/* Device-resident form: images, outputs and counts live in HBM.  d_imgs holds n_frames images,
 * frame f at d_imgs + f*frame_stride, rows `stride` bytes apart.  Fully asynchronous: every stage, the
 * quadtree included, is enqueued on the context's stream and the call returns at once; a capacity violation
 * detected on the device is reported (ORBGPU_E_CAPACITY) by the next orbgpu_synchronize / orbgpu_stage_times.
 * The images are read where they are (no copy) when stride == width rounded up to a multiple of 16 and d_imgs and
 * frame_stride are multiples of 16 (the bytes between width and stride may hold anything); any other layout is
 * copied into the library's frame slabs first.  Either way the images must stay untouched until the work this call
 * enqueued has completed (orbgpu_synchronize, or an event on orbgpu_stream); they are never written. */
int orbgpu_extract_batch_device(orbgpu_ctx *ctx, const uint8_t *d_imgs, int64_t frame_stride, int32_t stride,
                                int32_t n_frames, int32_t feature_count, orbgpu_keypoint *d_kp, uint8_t *d_desc,
                                int32_t capacity, int32_t *d_n_out);

/* Stage dumps of the LAST extract call, for stage-by-stage parity tests (host buffers, packed level
 * after level with row stride = level width; any pointer may be NULL).  cand_* receive the FAST
 * candidates of `frame` in reference order (coords relative to the 16-px border). */
int orbgpu_debug_dump(orbgpu_ctx *ctx, int32_t frame, uint8_t *pyramid, uint8_t *score, uint8_t *blurred,
                      int32_t *cand_x, int32_t *cand_y, int32_t *cand_score, int32_t cand_capacity,
                      int32_t *n_cand_per_level);

/* ---- matching -------------------------------------------------------------------------------- */
/* replaces: IFeatureExtractor::ComputeDistance (ifeature_extractor.h:48, orb_feature_extractor.cpp:83-98)
 * for host-side callers (map/map_point.cpp:152). */
static inline uint32_t orbgpu_hamming256(const uint8_t *a, const uint8_t *b) {
  uint32_t d = 0;
  for (int i = 0; i < 4; ++i) {
    uint64_t x, y;
    __builtin_memcpy(&x, a + 8 * i, 8);
    __builtin_memcpy(&y, b + 8 * i, 8);
    d += (uint32_t)__builtin_popcountll(x ^ y);
  }
  return d;
}

/* replaces: SNNMatcher<It>::MatchWithIterators (second_nearest_neighbor_matcher.hpp:30-80) with every
 * 'from' descriptor as candidate of every 'to' (no candidate restriction).
 * from_idx_per_to[i] = matched 'from' index or -1; dist[i] = its Hamming distance (undefined when -1). */
int orbgpu_match_bruteforce(orbgpu_ctx *ctx, const uint8_t *to_desc, int32_t n_to, const uint8_t *from_desc,
                            int32_t n_from, uint32_t threshold, double ratio, int32_t *from_idx_per_to,
                            uint32_t *dist);
int orbgpu_match_bruteforce_device(orbgpu_ctx *ctx, const uint8_t *d_to_desc, int32_t n_to,
                                   const uint8_t *d_from_desc, int32_t n_from, uint32_t threshold, double ratio,
                                   int32_t *d_from_idx_per_to, uint32_t *d_dist);

/* No counterpart in the reference; EXPLORATORY.  Selects how the matchers WITHOUT a candidate restriction --
 * orbgpu_match_bruteforce[_device] and orbgpu_match_allpairs[_angles]_device -- evaluate the distances:
 * ORBGPU_ENGINE_POPCOUNT (default, the contract path: XOR + v_bcnt per pair on the vector ALU) or ORBGPU_ENGINE_MFMA_I8
 * (descriptor bits expanded to +1 / -1 bytes, dot = 256 - 2 * distance exactly, on the i8 matrix cores; candidates above
 * the threshold are dropped a tile at a time).  Same results bit for bit; windowed / grouped matchers are not affected
 * (few candidates per row: nothing dense to contract), nor are brute-force calls of at most 4 Mi pairs.  With the MFMA
 * engine orbgpu_match_allpairs*_device takes at most 2880 keypoints per keyframe instead of 4080. */
#define ORBGPU_ENGINE_POPCOUNT 0
#define ORBGPU_ENGINE_MFMA_I8 1
int orbgpu_set_bruteforce_engine(orbgpu_ctx *ctx, int32_t engine);

/* No counterpart in the reference.  orbgpu_match_bruteforce_device for callers that cannot allocate device memory
 * (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc): same arguments, same results,
 * but its scratch (per-'from' claims and the per-chunk best / second partials) is a fixed 8 MiB __device__ array
 * inside the library instead of the context's growable buffer, and the call performs no hipMalloc / hipFree.
 * Limit: 8 * n_from (rounded up to 256) + 8 * n_to bytes must fit 8 MiB (ORBGPU_E_CAPACITY otherwise); the less room
 * is left, the fewer 'from' chunks run side by side.  Every call on a GPU shares that GPU's array -- from any context or
 * thread -- so calls that overlap on one GPU race: serialise them (one stream, or orbgpu_synchronize in between). */
int orbgpu_match_bruteforce_noalloc_device(orbgpu_ctx *ctx, const uint8_t *d_to_desc, int32_t n_to,
                                           const uint8_t *d_from_desc, int32_t n_from, uint32_t threshold, double ratio,
                                           int32_t *d_from_idx_per_to, uint32_t *d_dist);

/* replaces: the same matcher driven by AreaToIterator (iterators/area_to_iterator.h:16-58) or
 * ProjectionSearchIterator (iterators/projection_search_iterator.cpp:12-27): candidates of 'to' i are
 * the 'from' j (ascending j) with level in the range rule of Features::ListFeaturesInArea
 * (features.cpp:94,109-114) and |x_j - x_i| < window_i, |y_j - y_i| < window_i (strict, double).
 * to_window is the radius ALREADY truncated to an unsigned integer, as `const size_t&` does at
 * features.cpp:82.  to_active (optional) = 0 skips a 'to' row (AreaToIterator visits id 0 and every
 * level-0 id only).  From 2048 'from' rows on the library narrows the scan with the 64 x 48 search grid of
 * Features (features.cpp:127-161, laid over the context's image size; see orbgpu_match_window_grid_segments_device):
 * same results, 0.73 -> 0.07 ms for 5000 x 5000 keypoints of a 1080p frame with 15 px windows. */
typedef struct {
  const uint8_t *to_desc;      /* n_to x 32 */
  const double *to_xy;         /* n_to x 2 */
  const uint64_t *to_window;   /* n_to */
  const int32_t *to_min_level; /* n_to */
  const int32_t *to_max_level; /* n_to */
  const uint8_t *to_active;    /* n_to or NULL */
  int32_t n_to;
  const uint8_t *from_desc;    /* n_from x 32 */
  const double *from_xy;       /* n_from x 2 */
  const int32_t *from_level;   /* n_from */
  int32_t n_from;
  uint32_t threshold;
  double ratio;
} orbgpu_window_args;
int orbgpu_match_window(orbgpu_ctx *ctx, const orbgpu_window_args *args, int32_t *from_idx_per_to, uint32_t *dist);
int orbgpu_match_window_device(orbgpu_ctx *ctx, const orbgpu_window_args *device_args, int32_t *d_from_idx_per_to,
                               uint32_t *d_dist);

/* replaces: many SNNMatcher<ProjectionSearchIterator>::MatchWithIterators calls (one per frame: MonocularFrame::
 * SearchInVisiblePoints, frame/monocular/monocular_frame.cpp:285-306; MonocularKeyFrame::MatchVisibleMapPoints,
 * monocular_key_frame.cpp:183-216) in ONE launch, for offline batches: *device_args holds the arrays of all searches
 * back to back (device pointers; n_to / n_from = total rows), and segment s searches the 'to' rows
 * [to_begin, to_begin + to_count) among the 'from' rows [from_begin, from_begin + from_count) only -- same candidate
 * rule, threshold, ratio and per-'from' uniqueness as orbgpu_match_window, independently per segment.
 * d_from_idx_per_to[i] is the 'from' index RELATIVE to the segment's from_begin, or -1.  'to' rows outside every
 * segment are not written.  Segments must not overlap in 'to' rows; 'from' ranges may be shared only by segments
 * whose uniqueness is meant to be shared.  max_to_count = the largest to_count (the host knows it from building the
 * segments); d_segments has n_segments entries.  Asynchronous on the context's stream. */
typedef struct {
  int32_t to_begin, to_count, from_begin, from_count;
} orbgpu_segment;
int orbgpu_match_window_segments_device(orbgpu_ctx *ctx, const orbgpu_window_args *device_args,
                                        const orbgpu_segment *d_segments, int32_t n_segments, int32_t max_to_count,
                                        int32_t *d_from_idx_per_to, uint32_t *d_dist);

/* replaces: Features::AssignFeaturesToGrid / PosInGrid (features.cpp:127-161) -- the 64 x 48 search grid
 * (constants.h:18-19) the reference means to have: the body of PosInGrid BEHIND its early `return true` (:146-148;
 * as shipped every keypoint lands in grid[0][0]).  Range r = rows [begin, begin + count) of d_xy, given as
 * d_ranges[2r], d_ranges[2r + 1] (one range per frame); for cell c = pos_X * 48 + pos_Y (the flat layout of
 * grid[COLS][ROWS], features.h:36) the members are d_items[r * max_count + k], k in
 * [d_cell_start[r * 3073 + c], d_cell_start[r * 3073 + c + 1]), as row - begin in ascending order (push_back order).
 * max_count = the largest count (< 2^20).  Asynchronous on the context's stream. */
#define ORBGPU_GRID_COLS 64
#define ORBGPU_GRID_ROWS 48
int orbgpu_grid_build_device(orbgpu_ctx *ctx, const double *d_xy, const int32_t *d_ranges, int32_t n_ranges,
                             int32_t max_count, double image_width, double image_height, int32_t *d_cell_start,
                             int32_t *d_items);

/* replaces: orbgpu_match_window_segments_device's scan with Features::ListFeaturesInArea over that WORKING grid
 * (features.cpp:89-123: cells of [point - window, point + window] only).  Same arguments plus max_from_count (the largest
 * from_count) and the image size the grid is laid over (Features::image_width / image_height); same results bit for bit:
 * the grid narrows the scan, the window / level predicate still decides, and ties keep the smallest 'from' index.  The
 * library bins every segment's 'from' rows by cell and walks the 'to' rows along a Hilbert curve over the image, so one
 * wavefront's windows cover few cells.  Pays off when a window covers a small part of the image (1080p / 4K frames, search
 * by projection with radii of 15-30 px); at 640 x 480 / 1000 keypoints it is still about 3 x the plain scan.
 * max_to_count / max_from_count size the call's scratch rows: a segment that holds more than they say has its surplus
 * rows ignored and the next orbgpu_synchronize returns ORBGPU_E_CAPACITY (the same holds for max_count of
 * orbgpu_grid_build_device). */
int orbgpu_match_window_grid_segments_device(orbgpu_ctx *ctx, const orbgpu_window_args *device_args,
                                             const orbgpu_segment *d_segments, int32_t n_segments, int32_t max_to_count,
                                             int32_t max_from_count, double image_width, double image_height,
                                             int32_t *d_from_idx_per_to, uint32_t *d_dist);

/* replaces: the matcher inside DBoW2Handler::FastMatch (features/handlers/DBoW2/dbo_w2_handler.cpp:25-72):
 * SNNMatcher<BowToIterator>, candidates restricted to features of the same vocabulary node
 * (handlers/DBoW2/bow_to_iterator.cpp:12-47, bow_to_pointee.h:37-45).  The caller joins the two
 * DBoW2::FeatureVector maps (merge join on node id, as AdvanceUntilSameNode does) and passes the shared nodes
 * in node order: group g = 'to' ids to_ids[to_off[g] .. to_off[g+1]) and 'from' ids
 * from_ids[from_off[g] .. from_off[g+1]), each in the order of the node's std::vector.  A 'to' id may appear in
 * one group only (a feature has one node).  Thresholds of the three severities: dbo_w2_handler.cpp:37-51.
 * Outputs are indexed by 'to' id: from index or -1, and the distance.  The rotation filter is
 * orbgpu_rotation_filter, as in the reference (:68-71). */
typedef struct {
  const uint8_t *to_desc;   /* n_to x 32 */
  int32_t n_to;
  const uint8_t *from_desc; /* n_from x 32 */
  int32_t n_from;
  int32_t n_groups;
  const int32_t *to_off;    /* n_groups + 1, to_off[0] = 0 */
  const uint32_t *to_ids;   /* to_off[n_groups] */
  const int32_t *from_off;  /* n_groups + 1, from_off[0] = 0 */
  const uint32_t *from_ids; /* from_off[n_groups] */
  uint32_t threshold;
  double ratio;
} orbgpu_group_args;
int orbgpu_match_groups(orbgpu_ctx *ctx, const orbgpu_group_args *args, int32_t *from_idx_per_to, uint32_t *dist);
/* device form: every pointer in *device_args and both outputs are device pointers; n_slots = to_off[n_groups]
 * (the host knows it from building the join); asynchronous on the context's stream. */
int orbgpu_match_groups_device(orbgpu_ctx *ctx, const orbgpu_group_args *device_args, int32_t n_slots,
                               int32_t *d_from_idx_per_to, uint32_t *d_dist);

/* replaces: MonocularFrame::ComputeMatchesForLinking (frame/monocular/monocular_frame.cpp:173-189) for
 * consecutive frames of a device-resident batch produced by orbgpu_extract_batch_device: for pair p,
 * 'to' = frame to_frame[p], 'from' = frame from_frame[p]; AreaToIterator window semantics
 * (level 0 both sides), SNN(ratio, threshold), then, when check_orientation != 0, the
 * OrientationValidator filter.  d_match is n_pairs x capacity (from index or -1). */
int orbgpu_match_frames_device(orbgpu_ctx *ctx, const orbgpu_keypoint *d_kp, const uint8_t *d_desc,
                               const int32_t *d_n, int32_t capacity, const int32_t *d_to_frame,
                               const int32_t *d_from_frame, int32_t n_pairs, uint64_t window, uint32_t threshold,
                               double ratio, int32_t check_orientation, int32_t *d_match, uint32_t *d_dist,
                               int32_t *d_n_matches);

/* replaces: loop-candidate / keyframe-to-keyframe matching built on SNNMatcher<It>::MatchWithIterators
 * (features/matching/second_nearest_neighbor_matcher.hpp:30-80; call shape of MonocularKeyFrame::
 * FindMatchingMapPoints, frame/monocular/monocular_key_frame.cpp:277-301) in its all-pairs form:
 * every keyframe r of the row set ('to') against every keyframe c of the column set ('from'), every 'from'
 * keypoint a candidate in index order, then ratio / threshold / uniqueness and, when check_orientation != 0,
 * OrientationValidator::Validate (validators/orientation_validator.cpp:16-97).
 * Both sets are laid out like the output of orbgpu_extract_batch_device: keyframe k occupies rows
 * [k*capacity, k*capacity + n[k]) of its descriptor (x32 bytes) and keypoint arrays; capacity <= 4080.
 * row_offset >= 0 names the column index of row 0 when the rows are a slice of the columns (one rank's shard of
 * an all-gathered set): the pair (r, row_offset + r) is then skipped and reports 0 matches; pass -1 otherwise.
 * d_n_matches is n_rows x n_cols; d_match (n_rows x n_cols x capacity, 'from' index or -1) may be NULL.
 * Asynchronous on the context's stream; all pointers are device pointers. */
int orbgpu_match_allpairs_device(orbgpu_ctx *ctx, const orbgpu_keypoint *d_kp_rows, const uint8_t *d_desc_rows,
                                 const int32_t *d_n_rows, int32_t n_rows, const orbgpu_keypoint *d_kp_cols,
                                 const uint8_t *d_desc_cols, const int32_t *d_n_cols, int32_t n_cols, int32_t capacity,
                                 int32_t row_offset, uint32_t threshold, double ratio, int32_t check_orientation,
                                 int32_t *d_n_matches, int32_t *d_match);

/* Same matcher (same reference counterpart: second_nearest_neighbor_matcher.hpp:30-80 + orientation_validator.cpp:16-97)
 * for keyframe sets that travelled between GPUs: the rotation filter reads nothing of a keypoint but its angle, so the
 * sharded workload all-gathers descriptors + a float side array (4 B per keypoint) instead of the 32-byte
 * records.  d_angle_rows / d_angle_cols are [n][capacity] floats = KeyPoint::angle as orbgpu_keypoint holds it. */
int orbgpu_match_allpairs_angles_device(orbgpu_ctx *ctx, const float *d_angle_rows, const uint8_t *d_desc_rows,
                                        const int32_t *d_n_rows, int32_t n_rows, const float *d_angle_cols,
                                        const uint8_t *d_desc_cols, const int32_t *d_n_cols, int32_t n_cols,
                                        int32_t capacity, int32_t row_offset, uint32_t threshold, double ratio,
                                        int32_t check_orientation, int32_t *d_n_matches, int32_t *d_match);

/* replaces: OrientationValidator::Validate (validators/orientation_validator.cpp:16-97).
 * match[i] >= 0 means 'to' i is matched to 'from' match[i]; discarded matches are set to -1. */
int orbgpu_rotation_filter(orbgpu_ctx *ctx, const float *angle_to, int32_t n_to, const float *angle_from,
                           int32_t n_from, int32_t *match, int32_t *n_discarded);

/* replaces: MapPoint::ComputeDistinctiveDescriptor (map/map_point.cpp:135-171) for a batch of map points.
 * The observed descriptors of point p are rows offsets[p] .. offsets[p+1] of desc (the order in which
 * Observation::AppendDescriptorsToList appended them).  best_index[p] = position inside the group of the
 * descriptor with the smallest median distance to the group (median = element (size_t)(0.5*(N-1)) of the
 * sorted row, first row on ties), or -1 for an empty group (the reference leaves such a point untouched).
 * A group may hold at most 65535 descriptors. */
int orbgpu_distinctive_descriptor(orbgpu_ctx *ctx, const uint8_t *desc, const int32_t *offsets, int32_t n_points,
                                  int32_t *best_index);
int orbgpu_distinctive_descriptor_device(orbgpu_ctx *ctx, const uint8_t *d_desc, const int32_t *d_offsets,
                                         int32_t n_points, int32_t *d_best_index);

/* ---- per-keypoint camera geometry ----------------------------------------------------------------- */
/* replaces: the loop of FeatureHandlerFactory::Create (factories/feature_handler_factory.cpp:53-60) over
 * MonocularCamera::UndistortPoint (camera/monocular_camera.cpp:49-61) and ::UnprojectAndUndistort (:40-47).
 * model: ORBGPU_DISTORTION_NONE (no distortion model set), _BARREL5 (k = k1 k2 p1 p2 k3, barrel5.cpp:44-70),
 * _FISHEYE (k = k1 k2 k3 k4, fish_eye.cpp:68-116) -- the values of camera/distortions/distortion_model_type.h;
 * BARREL8 is rejected (the reference throws "not yet implemented", barrel8.cpp:52-55).
 * undistorted: 2 doubles per keypoint, unprojected: 3 doubles per keypoint, ok: bit 0 / bit 1 = the bool the two
 * reference calls return.  Where UndistortPoint returns false the reference leaves its output slot untouched
 * (uninitialised after resize); this library writes (0, 0).  Barrel5 results are bit-exact; FishEye goes through
 * tan() and agrees with a glibc build to a few ulp. */
#define ORBGPU_DISTORTION_NONE (-1)
#define ORBGPU_DISTORTION_BARREL5 0
#define ORBGPU_DISTORTION_BARREL8 1
#define ORBGPU_DISTORTION_FISHEYE 2
typedef struct {
  double fx, fy, cx, cy;
  int32_t model;
  int32_t reserved;
  double k[8];
} orbgpu_camera;
int orbgpu_keypoint_geometry(orbgpu_ctx *ctx, const orbgpu_camera *camera, const orbgpu_keypoint *kp, int32_t n,
                             double *undistorted, double *unprojected, uint8_t *ok);
/* device form on the [n_frames][capacity] output of orbgpu_extract_batch_device (d_n = its per-frame counts);
 * outputs use the same [frame][capacity] slots; asynchronous on the context's stream. */
int orbgpu_keypoint_geometry_device(orbgpu_ctx *ctx, const orbgpu_camera *camera, const orbgpu_keypoint *d_kp,
                                    const int32_t *d_n, int32_t n_frames, int32_t capacity, double *d_undistorted,
                                    double *d_unprojected, uint8_t *d_ok);

/* ---- vocabulary (bag of words) ------------------------------------------------------------------ */
/* replaces: the per-feature tree descent of DBoW2Handler::Precompute (features/handlers/DBoW2/
 * dbo_w2_handler.cpp:74-93) = TemplatedVocabulary<FORB>::transform(feature, word_id, weight, nid, levelsup)
 * (DBoW2/DBoW2/TemplatedVocabulary.h:1218-1259) with FORB::distance (DBoW2/DBoW2/FORB.cpp:81-101).
 * The tree is passed flat, as m_nodes holds it: node 0 is the root, the children of node v are
 * children[child_off[v] .. child_off[v+1]) in Node::children order (a leaf has none), node_desc is
 * n_nodes x 32 bytes, word_id[v] = Node::word_id (read for leaves only), levels = m_L.  The arrays are copied
 * to the context's device; the handle must be destroyed before the context.  Weights, the BowVector
 * accumulation / normalisation and FeatureVector::addFeature stay on the host (TemplatedVocabulary.h:1125-1189;
 * see orb_slam3_amd/host/hip_bow_vocabulary.h): they are map operations in double on a few hundred words. */
typedef struct orbgpu_vocab orbgpu_vocab;
int orbgpu_vocab_create(orbgpu_ctx *ctx, int32_t n_nodes, const int32_t *child_off, const uint32_t *children,
                        const uint8_t *node_desc, const uint32_t *word_id, int32_t levels, orbgpu_vocab **out);
void orbgpu_vocab_destroy(orbgpu_vocab *vocab);
/* word_id_out[i] = leaf word of descriptor i, node_id_out[i] = its ancestor at level (levels - levelsup), the
 * root (0) when that level is <= 0.  Host pointers; the _device form takes device pointers and is asynchronous. */
int orbgpu_bow_transform(orbgpu_ctx *ctx, const orbgpu_vocab *vocab, const uint8_t *desc, int32_t n, int32_t levelsup,
                         uint32_t *word_id_out, uint32_t *node_id_out);
int orbgpu_bow_transform_device(orbgpu_ctx *ctx, const orbgpu_vocab *vocab, const uint8_t *d_desc, int32_t n,
                                int32_t levelsup, uint32_t *d_word_id_out, uint32_t *d_node_id_out);

/* ---- keyframe database (place recognition) --------------------------------------------------------- */
/* replaces: DBoW2Database::SearchWordSharingKeyFrames (frame/database/DBoW2/dbo_w2_database.cpp:83-103) and the
 * vocabulary->score(query BowVector, keyframe BowVector) calls of FilterCandidatesByWordSharingAcceptableScore
 * (:114-132) = {L1,L2,ChiSquare,Bhattacharyya,DotProduct}Scoring::score (DBoW2/DBoW2/ScoringObject.cpp:23-120,
 * :125-170, :226-311), for ONE query against ALL keyframes of the database in one call.
 * The database is passed keyframe-major (the transpose of inverted_file_, dbo_w2_database.h:55): the words of
 * keyframe k are entries kf_offsets[k] .. kf_offsets[k+1]) of kf_words (strictly ascending inside a keyframe, the
 * std::map order of its BowVector), kf_values their WordValue, kf_counts the inverted-file value of (word,
 * keyframe) (= DBoW2Handler::GetWordFrequencies, dbo_w2_handler.cpp:90-92; NULL = all 1).  The query is given the
 * same way (q_words strictly ascending).
 * shared_words[k] = sum over common words of kf_count * q_count (0 = the keyframe is not in the reference's
 * WordSharingKeyFrameMap); score[k] = score(query, keyframe k), accumulated in word order, bit-exact in double
 * (the reference evaluates it only where shared_words passes the 0.8 * max cut; here every keyframe gets one).
 * scoring = DBoW2::ScoringType (BowVector.h): 0 L1_NORM, 1 L2_NORM, 2 CHI_SQUARE, 4 BHATTACHARYYA, 5 DOT_PRODUCT;
 * 3 (KL) is rejected with ORBGPU_E_INVALID. */
typedef struct {
  const uint32_t *q_words;
  const double *q_values;
  const uint32_t *q_counts;
  int32_t n_q;
  int32_t n_keyframes;
  const int64_t *kf_offsets;
  const uint32_t *kf_words;
  const double *kf_values;
  const uint32_t *kf_counts;
  int32_t scoring;
  int32_t reserved;
} orbgpu_bow_score_args;
int orbgpu_bow_score(orbgpu_ctx *ctx, const orbgpu_bow_score_args *args, uint64_t *shared_words, double *score);
/* device form: every pointer of the struct and both outputs are device pointers (a database that stays resident
 * between queries); asynchronous on the context's stream; the orderings are the caller's responsibility. */
int orbgpu_bow_score_device(orbgpu_ctx *ctx, const orbgpu_bow_score_args *device_args, uint64_t *d_shared_words,
                            double *d_score);

/* A database that stays resident on the device between queries -- what IKeyFrameDatabase
 * (frame/database/ikey_frame_database.h:14-29) is to its callers: Append (dbo_w2_database.cpp:53-61), Erase
 * (:187-195) and the query part of DetectNBestCandidates (:63-81).  A keyframe is named by a caller-chosen 64-bit id
 * (KeyFrame::Id(), the value Serialize writes, :46).  words / values = the keyframe's BowVector in std::map order
 * (strictly ascending), counts = its word frequencies (NULL = 1 per word; a database created with with_counts = 0
 * stores none and accepts only NULL).  Appending an id that is present replaces its entry.  Erasing an absent id is
 * not an error (the reference's map::erase is a no-op).  The handle must be destroyed before its context. */
typedef struct orbgpu_bowdb orbgpu_bowdb;
int orbgpu_bowdb_create(orbgpu_ctx *ctx, int32_t scoring, int32_t with_counts, orbgpu_bowdb **out);
void orbgpu_bowdb_destroy(orbgpu_bowdb *db);
int orbgpu_bowdb_append(orbgpu_bowdb *db, uint64_t keyframe_id, const uint32_t *words, const double *values,
                        const uint32_t *counts, int32_t n_words);
int orbgpu_bowdb_erase(orbgpu_bowdb *db, uint64_t keyframe_id);
int orbgpu_bowdb_size(const orbgpu_bowdb *db, int32_t *n_keyframes);
/* One query against every resident keyframe (host pointers): keyframe_ids / shared_words / score receive
 * *n_out <= capacity entries in the database's internal order; ORBGPU_E_CAPACITY when capacity < size. */
int orbgpu_bowdb_query(orbgpu_bowdb *db, const uint32_t *q_words, const double *q_values, const uint32_t *q_counts,
                       int32_t n_q, int32_t capacity, uint64_t *keyframe_ids, uint64_t *shared_words, double *score,
                       int32_t *n_out);

/* ---- measurement ----------------------------------------------------------------------------- */
/* Per-stage GPU time in milliseconds (HIP events on the context's stream), averaged over the extract calls
 * enqueued since the last orbgpu_synchronize / orbgpu_stage_times (at most the last 32); waits for the stream.
 * ms8 = 8 floats: [0] pyramid (the resize chain; level 0's FAST launch runs beside it on the second stream)
 * [1] FAST score + cell NMS / fallback / compaction (one fused kernel) of the levels >= 1 -- the launch on the critical
 * path, the one the roofline is quoted on [2] orientation (its kernel runs behind the quadtree and beside the blur, off
 * the critical path) [3] quadtree [4] blur [5] descriptor [6] total [7] the same FAST kernel's launch for level 0 (it
 * needs no pyramid; 0 when ORBGPU_EARLY_L0=0 puts all levels into launch [1]).  With more than one slice per batch
 * (orbgpu_set_batch_slices) [0]..[5] and [7] are the kernels of slice 0 -- its share of the frames, running beside the
 * other slices -- and [6] still spans the whole batch.  Used by bench.py for the roofline line. */
int orbgpu_stage_times(orbgpu_ctx *ctx, float *ms8);
/* Which stage boundaries an extract call records.  An event between two kernels costs 6-7 us of stream time, so:
 * 0 (the default): only the FAST kernel's two ends (the kernel the roofline is quoted on) -- orbgpu_stage_times fills
 * [1] and leaves the rest 0; time the whole call from the host.  1: every boundary and the call's two ends (about 3 %
 * of a 256-frame 640x480 batch), which is why it is a switch. */
int orbgpu_set_stage_timing(orbgpu_ctx *ctx, int32_t all_stages);
/* Runs a register-resident XOR + v_bcnt_u32_b32 loop and returns descriptor pairs per second: the
 * measured popcount ceiling that BASELINE.md asks to report next to the nominal one. */
int orbgpu_popcount_peak(orbgpu_ctx *ctx, double *pairs_per_second);

#ifdef __cplusplus
}
#endif
#endif /* ORBGPU_H_ */
