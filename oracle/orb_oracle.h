/*
 * orb_oracle.h -- CPU restatement of the reference ORB front-end (TEST INFRASTRUCTURE ONLY).
 *
 * This is the parity oracle for the HIP path.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it.  The shipped library (orb_slam3_amd/csrc) never
 * links, includes or calls anything in this directory.
 *
 * PARITY STATUS: the reference (vahagnIV/orb_slam3) cannot be built in this image (needs
 * OpenCV, Eigen, g2o, Boost, spdlog, gtest -- all absent), so no
 * oracle/_ref build exists.  The extraction half (pyramid, FAST, quadtree, orientation, blur,
 * descriptor) restates OpenCV 4.5.4 generic-path arithmetic from its published algorithm and
 * is "parity unpinned" (the reference's tests hold no golden vectors for it).  The window
 * query is pinned by the reference's own known-answer test
 * (test/features/feature_tests.cpp:21-50); Hamming / SNN / rotation-histogram are fully defined
 * by reference source and checked against brute-force restatements in tests/.
 *
 * All citations are file:line under /root/reference/.
 */
#ifndef ORB_ORACLE_H_
#define ORB_ORACLE_H_

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OO_MAX_LEVELS 16
#define OO_HALF_PATCH 15   /* src/features/orb_feature_extractor.h:18 */
#define OO_PATCH 31        /* :19 */
#define OO_EDGE 19         /* :20 */

typedef struct {
  int width, height;      /* ctor args, orb_feature_extractor.h:24-29 */
  double scale_factor;
  int levels;
  int fast_init, fast_min;
} oo_params;

/* features::KeyPoint, src/features/key_point.h:10-23 */
typedef struct {
  double x, y;
  double angle; /* holds a float value (degrees) */
  double size;
  int32_t level;
  int32_t pad_;
} oo_keypoint;

/* one FAST candidate handed to the quadtree (coords relative to minBorder = 16) */
typedef struct {
  float x, y, response;
} oo_cand;

typedef struct {
  int levels;
  int w[OO_MAX_LEVELS], h[OO_MAX_LEVELS];
  int budget[OO_MAX_LEVELS];
  int ncols[OO_MAX_LEVELS], nrows[OO_MAX_LEVELS], wcell[OO_MAX_LEVELS], hcell[OO_MAX_LEVELS];
  int nini[OO_MAX_LEVELS];
  int kp_size[OO_MAX_LEVELS];
  double scale[OO_MAX_LEVELS], inv_scale[OO_MAX_LEVELS], sigma2[OO_MAX_LEVELS];
} oo_geometry;

/* ---- geometry (orb_feature_extractor.cpp:53-69, 211-242, 284, 310, 506-508) ---- */
int oo_geometry_compute(const oo_params *p, int n_features, oo_geometry *g);

/* ---- primitives ---- */
/* cv::resize(..., INTER_LINEAR) for CV_8UC1, call site orb_feature_extractor.cpp:517 */
int oo_resize_linear_u8(const uint8_t *src, int sw, int sh, int sstride, uint8_t *dst, int dw, int dh,
                        int dstride);
/* cv::GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101), call site :559 (App. A.3) */
void oo_gaussian_blur7(const uint8_t *src, int w, int h, int sstride, uint8_t *dst, int dstride);
/* cv::fastAtan2, call site :201 (App. A.4) */
float oo_fast_atan2(float y, float x);
/* IC_Angle :177-202 */
float oo_ic_angle(const uint8_t *img, int stride, int cx, int cy);
/* computeOrbDescriptor :120-165; angle_deg is kpt.angle */
void oo_orb_descriptor(const uint8_t *blurred, int stride, int cx, int cy, double angle_deg, uint8_t *desc32);
/* cos / sin variant of the descriptor rotation: 0 = cos(double) rounded to float (default), 1 = cosf / sinf of the host libm */
void oo_set_trig_variant(int variant);
int oo_get_trig_variant(void);
void oo_descriptor_trig(double angle_deg, int variant, float *cos_out, float *sin_out);
/* closed-form FAST-9/16 score  max(M_dark, M_bright) - 1  (App. A.2); may be <= 0 */
int oo_fast_score(const uint8_t *img, int stride, int x, int y);
/* threshold-independent score map: S = score if score >= min_thr else 0, zero outside [3,w-4]x[3,h-4] */
void oo_fast_score_map(const uint8_t *img, int w, int h, int stride, int min_thr, uint8_t *score, int ostride);
/* literal cv::FAST(roi, kps, thr, nonmax=true) on a cw x ch ROI (row-buffer algorithm). Returns count. */
int oo_fast_literal(const uint8_t *roi, int stride, int cw, int ch, int thr, int nonmax, int cap, int *ox,
                    int *oy, int *oscore);
/* ComputeKeyPointsOctTree :227-276 for one level: cell loop + fallback; candidates in reference order.
 * use_literal=1 runs oo_fast_literal per cell, 0 uses the score map + per-cell NMS. Returns count or <0. */
int oo_detect_level(const uint8_t *img, int w, int h, int stride, int fast_init, int fast_min, int use_literal,
                    oo_cand *out, int cap);
/* DistributeOctTree :300-502 (+DivideNode :602-652). Returns number of keypoints written (<= cap) or <0.
 * out_x/out_y are still relative to minBorder.  *tie_ambiguous gets 1 when the allocator-dependent
 * std::sort tie (App. B.3) straddled the stop point, i.e. the reference output is not a pure function
 * of the image for this call. */
int oo_distribute_octtree(const oo_cand *cands, int n, int minX, int maxX, int minY, int maxY, int n_features,
                          float *out_x, float *out_y, float *out_resp, int cap, int *tie_ambiguous);

/* ---- full extraction, Extract :530-600 ---- */
typedef struct {
  /* optional dumps (may be NULL); pyramid/blur/score are packed level after level, row stride = w[l] */
  uint8_t *pyramid;
  uint8_t *blurred;
  uint8_t *score;
  oo_cand *cands;      /* capacity cands_cap, all levels concatenated */
  int cands_cap;
  int n_cands[OO_MAX_LEVELS];
  int n_kp[OO_MAX_LEVELS];
  int tie_ambiguous[OO_MAX_LEVELS];
} oo_dump;

/* returns number of keypoints (reference returns monoIndex=0; we return the count), -1 on empty image,
 * <-1 on unsupported geometry. kps/desc have capacity cap. Output order = reference order (reversed, B.4). */
int oo_extract(const oo_params *p, const uint8_t *img, int stride, int n_features, oo_keypoint *kps,
               uint8_t *desc, int cap, oo_dump *dump);

/* identical results; FAST evaluated with the literal per-cell cv::FAST (early rejection) -- the CPU baseline */
int oo_extract_literal(const oo_params *p, const uint8_t *img, int stride, int n_features, oo_keypoint *kps,
                       uint8_t *desc, int cap);

/* ---- matching ---- */
/* ComputeDistance :83-98 (bit-hack popcount) */
unsigned oo_hamming256(const uint8_t *a, const uint8_t *b);
/* Features::ListFeaturesInArea features.cpp:81-125 (degenerate grid -> linear scan) */
int oo_list_features_in_area(const oo_keypoint *kps, int n, double px, double py, size_t window, int min_level,
                             int max_level, int32_t *out_idx);

/* the 64 x 48 search grid behind PosInGrid's early return (features.cpp:127-161, constants.h:18-19) */
#define OO_GRID_COLS 64
#define OO_GRID_ROWS 48
void oo_pos_in_grid(double kx, double ky, double image_width, double image_height, int *posX, int *posY);
void oo_assign_features_to_grid(const double *xy, int n, double image_width, double image_height,
                                int32_t *cell_start, int32_t *items);
int oo_list_features_in_area_grid(const double *xy, const int32_t *level, const int32_t *cell_start,
                                  const int32_t *items, double image_width, double image_height, double px, double py,
                                  size_t window, int min_level, int max_level, int32_t *out_idx);

/* SNNMatcher::MatchWithIterators second_nearest_neighbor_matcher.hpp:30-80.
 * Generic form: for 'to' i (in order, skipping to_active[i]==0 when given) candidates are the 'from' j
 * (ascending) that pass the window predicate (when to_xy != NULL) -- see oo_list_features_in_area.
 * match_from[i] = j or -1; match_dist[i] = best distance (valid where match_from>=0). */
typedef struct {
  const uint8_t *to_desc; int n_to;
  const uint8_t *from_desc; int n_from;
  /* window predicate; to_xy == NULL -> brute force over all from */
  const double *to_xy;         /* n_to x 2 */
  const uint64_t *to_window;   /* n_to, already truncated to size_t as the reference does (features.cpp:82) */
  const int32_t *to_min_level; /* n_to */
  const int32_t *to_max_level; /* n_to */
  const uint8_t *to_active;    /* optional n_to */
  const double *from_xy;       /* n_from x 2 */
  const int32_t *from_level;   /* n_from */
  unsigned threshold;
  double ratio;
} oo_match_args;
int oo_snn_match(const oo_match_args *a, int32_t *match_from, uint32_t *match_dist);

/* AreaToIterator semantics (iterators/area_to_iterator.h:24-43, area_to_pointee.h:43-54): fills
 * to_active (id 0 always, id>0 iff level==0), window=r, levels [0,0]. */
int oo_match_area(const oo_keypoint *kp_to, const uint8_t *desc_to, int n_to, const oo_keypoint *kp_from,
                  const uint8_t *desc_from, int n_from, size_t window, unsigned threshold, double ratio,
                  int32_t *match_from, uint32_t *match_dist);

/* Post-extract per-keypoint geometry, factories/feature_handler_factory.cpp:53-60:
 *   MonocularCamera::UndistortPoint        camera/monocular_camera.cpp:49-61   -> undist[2i .. 2i+1]
 *   MonocularCamera::UnprojectAndUndistort camera/monocular_camera.cpp:40-47   -> unproj[3i .. 3i+2]
 * with UnprojectPoint / ProjectPoint (:31-38), fx_inv_ = fx ? 1/fx : 0, fy_inv_ = fy ? 1/fy : 1
 * (monocular_camera.h:107-114), Barrel5::UnDistortPoint (camera/distortions/barrel5.cpp:44-70) and
 * FishEye::UnDistortPoint (camera/distortions/fish_eye.cpp:68-116).  model: -1 no distortion model, 0 BARREL5
 * (k = k1 k2 p1 p2 k3), 2 FISHEYE (k = k1 k2 k3 k4) -- distortion_model_type.h.  ok[i] bit 0 / bit 1 = the bool the
 * two calls return.  Where UndistortPoint fails the reference leaves its output slot untouched (uninitialised
 * memory after resize); this restatement writes (0, 0) there. */
typedef struct {
  double fx, fy, cx, cy;
  int32_t model;
  int32_t reserved;
  double k[8];
} oo_camera;
void oo_keypoint_geometry(const oo_keypoint *kp, int n, const oo_camera *cam, double *undist, double *unproj, uint8_t *ok);

/* MapPoint::ComputeDistinctiveDescriptor (map/map_point.cpp:135-171) for a batch of map points: the descriptors
 * of point p are rows offsets[p] .. offsets[p+1] of desc.  best[p] = index (inside the group) of the descriptor
 * whose sorted distance row has the smallest element at position (size_t)(0.5 * (N - 1)); the first such row
 * wins (strict <, :162); -1 for an empty group (:146-148 leaves the point untouched). */
void oo_distinctive_descriptor(const uint8_t *desc, const int32_t *offsets, int n_points, int32_t *best);

/* TemplatedVocabulary<FORB>::transform for one feature (DBoW2/DBoW2/TemplatedVocabulary.h:1218-1259) with
 * FORB::distance (DBoW2/DBoW2/FORB.cpp:81-101), for every row of desc.  The tree is given flat: the children of
 * node v are children[child_off[v] .. child_off[v+1]) in the order of Node::children (a leaf has none), node 0 is
 * the root, node_desc is n_nodes x 32, word_id[v] is Node::word_id of a leaf.  out_node receives the ancestor at
 * level L - levelsup (0 = root when that level is <= 0), as transform's nid. */
void oo_bow_transform(const uint8_t *desc, int n, const int32_t *child_off, const uint32_t *children,
                      const uint8_t *node_desc, const uint32_t *word_id, int L, int levelsup, uint32_t *out_word,
                      uint32_t *out_node);

/* DBoW2Handler::FastMatch candidate generation (handlers/DBoW2/dbo_w2_handler.cpp:25-72) with
 * BowToIterator / BowToPointee (handlers/DBoW2/bow_to_iterator.{h,cpp}, bow_to_pointee.h:19-59): both feature
 * vectors (DBoW2::FeatureVector = std::map<NodeId, std::vector<unsigned>>) are given in CSR form -- node ids
 * ascending, ids of node k in ids[off[k] .. off[k+1]).  'to' ids are visited node by node over the nodes the two
 * vectors share (merge join with lower_bound, bow_to_iterator.cpp:27-47); the candidates of a 'to' id are the
 * 'from' ids of the same node in list order; then SNNMatcher::MatchWithIterators.  Returns the match count. */
int oo_match_bow(const uint8_t *desc_to, int n_to, const uint8_t *desc_from, int n_from, const uint32_t *to_nodes,
                 const int32_t *to_off, const uint32_t *to_ids, int n_to_nodes, const uint32_t *from_nodes,
                 const int32_t *from_off, const uint32_t *from_ids, int n_from_nodes, unsigned threshold, double ratio,
                 int32_t *match_from, uint32_t *match_dist);

/* Keyframe-database query.  The database is passed keyframe by keyframe (words of keyframe k = entries kf_off[k] ..
 * kf_off[k+1]), ascending), exactly what DBoW2Database::Append receives per keyframe; the function first builds the
 * inverted file as Append does (frame/database/DBoW2/dbo_w2_database.cpp:53-61: inverted_file_[word][keyframe] =
 * count), then runs SearchWordSharingKeyFrames (:83-103) over the query's word frequencies and, for every keyframe,
 * vocabulary->score(query, keyframe) as FilterCandidatesByWordSharingAcceptableScore calls it (:127-128), restating
 * DBoW2/DBoW2/ScoringObject.cpp (L1 :23-68, L2 :73-120, chi-square :125-170, Bhattacharyya :226-266, dot product
 * :271-311; scoring = DBoW2::ScoringType, KL = 3 is not restated and yields NaN).  counts may be NULL (= 1). */
void oo_bow_score(const uint32_t *q_words, const double *q_values, const uint32_t *q_counts, int n_q,
                  const int64_t *kf_off, const uint32_t *kf_words, const double *kf_values, const uint32_t *kf_counts,
                  int n_kf, int scoring, uint64_t *shared_words, double *score);

/* OrientationValidator::Validate validators/orientation_validator.cpp:16-97.
 * match_from[i] >= 0 are the matches (to i -> from match_from[i]); discarded ones are set to -1.
 * returns number discarded. */
int oo_orientation_validate(const double *angle_to, const double *angle_from, int32_t *match_from, int n_to);

/* scalar helpers orb_feature_extractor.cpp:654-660, 854-873 */
unsigned oo_predict_scale(double distance, double max_distance, double scale_factor, int levels);

#ifdef __cplusplus
}
#endif
#endif
