// GPU-backed bodies for the reference's matcher entry points.
//
// The workhorse behind the same-name replacements in features/ (second_nearest_neighbor_matcher.hpp,
// validators/orientation_validator.h): SNNMatcher<It>::MatchWithIterators
// (src/features/matching/second_nearest_neighbor_matcher.hpp:30-80) per candidate generator, each gathering
// (descriptor, position, radius, level-range) arrays for ONE batched C-ABI call:
//   * MatchArea       <- SNNMatcher<AreaToIterator>          (frame/monocular/monocular_frame.cpp:173-184)
//   * MatchProjection <- SNNMatcher<ProjectionSearchIterator> (monocular_frame.cpp:285-306, monocular_key_frame.cpp:183-216,353-376)
//   * MatchBow        <- SNNMatcher<BowToIterator>          (features/handlers/DBoW2/dbo_w2_handler.cpp:25-72)
//   * HipOrientationValidator::Validate <- OrientationValidator::Validate (validators/orientation_validator.cpp:16-40)
// Compiled inside the reference tree (includes its headers).  Results are the same unordered maps.
#ifndef ORB_SLAM3_AMD_HIP_SNN_MATCHER_H_
#define ORB_SLAM3_AMD_HIP_SNN_MATCHER_H_

#include <algorithm>
#include <iterator>
#include <list>
#include <map>
#include <stdexcept>
#include <unordered_map>
#include <vector>

#include <features/features.h>                    // reference headers
#include <frame/map_point_visibility_params.h>
#include <map/map_point.h>
#include "orbgpu.h"

namespace orb_slam3 {
namespace features {
namespace matching {

class HipSNNMatcher {
 public:
  // same arguments as SNNMatcher(nearest_neighbour_ratio, threshold), second_nearest_neighbor_matcher.hpp:26-28
  HipSNNMatcher(orbgpu_ctx * ctx, precision_t nearest_neighbour_ratio, unsigned threshold)
      : ctx_(ctx), nearest_neighbour_ratio_(nearest_neighbour_ratio), THRESHOLD_(threshold) {}

  // AreaToIterator(begin_id .. end_id, to, from, window_size): level-0 window search on both sides; the first id is
  // visited whatever its level, the following ones only when they are level-0 keypoints
  // (iterators/area_to_iterator.h:24-43, area_to_pointee.h:43-54).  The reference calls it with 0 .. Size().
  void MatchArea(const Features & to, const Features & from, size_t window_size,
                 std::unordered_map<std::size_t, std::size_t> & out_matches, size_t begin_id = 0,
                 size_t end_id = (size_t)-1) const {
    const size_t nt = to.Size(), nf = from.Size();
    end_id = std::min(end_id, nt);
    std::vector<double> to_xy(2 * nt), from_xy(2 * nf);
    std::vector<uint64_t> win(nt, (uint64_t)window_size);
    std::vector<int32_t> lmin(nt, 0), lmax(nt, 0), flev(nf);
    std::vector<uint8_t> active(nt);
    for (size_t i = 0; i < nt; ++i) {
      to_xy[2 * i] = to.keypoints[i].X();
      to_xy[2 * i + 1] = to.keypoints[i].Y();
      active[i] = i >= begin_id && i < end_id && (i == begin_id || to.keypoints[i].level == 0);
    }
    Fill(from, from_xy, flev);
    Run(to.descriptors.data(), nt, to_xy, win, lmin, lmax, &active, from, from_xy, flev,
        [&](size_t i, size_t j) { out_matches[i] = j; });
  }

  // ProjectionSearchIterator(begin, end, from_features): 'to' = visible map points
  // (iterators/projection_search_pointee.cpp:20-35); window_size is truncated by the const size_t& parameter of
  // ListFeaturesInArea (features.cpp:82); levels [max(l-1,0), l+1]
  void MatchProjection(const std::list<frame::MapPointVisibilityParams> & visibles, const Features & from,
                       std::unordered_map<map::MapPoint *, std::size_t> & out_matches) const {
    MatchProjection(visibles.begin(), visibles.end(), from, out_matches);
  }
  void MatchProjection(std::list<frame::MapPointVisibilityParams>::const_iterator vis_begin,
                       std::list<frame::MapPointVisibilityParams>::const_iterator vis_end, const Features & from,
                       std::unordered_map<map::MapPoint *, std::size_t> & out_matches) const {
    const size_t nt = (size_t)std::distance(vis_begin, vis_end), nf = from.Size();
    std::vector<uint8_t> to_desc(32 * nt);
    std::vector<double> to_xy(2 * nt), from_xy(2 * nf);
    std::vector<uint64_t> win(nt);
    std::vector<int32_t> lmin(nt), lmax(nt), flev(nf);
    std::vector<map::MapPoint *> ids(nt);
    size_t i = 0;
    for (auto vit = vis_begin; vit != vis_end; ++vit) {
      const frame::MapPointVisibilityParams & v = *vit;
      const DescriptorType d = v.map_point->GetDescriptor();
      std::copy(d.data(), d.data() + 32, to_desc.begin() + 32 * i);
      to_xy[2 * i] = v.position.x();
      to_xy[2 * i + 1] = v.position.y();
      win[i] = (uint64_t)(size_t)v.window_size;
      lmin[i] = std::max(v.level - 1, 0);
      lmax[i] = v.level + 1;
      ids[i] = v.map_point;
      ++i;
    }
    Fill(from, from_xy, flev);
    Run(to_desc.data(), nt, to_xy, win, lmin, lmax, nullptr, from, from_xy, flev,
        [&](size_t a, size_t j) { out_matches[ids[a]] = j; });
  }

  // DBoW2Handler::FastMatch's matcher: candidates of a 'to' feature are the 'from' features of the same vocabulary
  // node.  FeatureVector is DBoW2::FeatureVector (std::map<NodeId, std::vector<unsigned>>) or anything shaped like
  // it; the merge join below is BowToIterator::AdvanceUntilSameNode (handlers/DBoW2/bow_to_iterator.cpp:27-47).
  template <class FeatureVector>
  void MatchBow(const FeatureVector & feature_vector_to, const FeatureVector & feature_vector_from, const Features & to,
                const Features & from, std::unordered_map<std::size_t, std::size_t> & out_matches) const {
    MatchBow(feature_vector_to, feature_vector_from, to, from, out_matches, feature_vector_to.begin(), feature_vector_to.end());
  }
  // the nodes [to_first, to_last) of feature_vector_to (BowToIterator begin / end, dbo_w2_handler.cpp:53-63)
  template <class FeatureVector>
  void MatchBow(const FeatureVector & feature_vector_to, const FeatureVector & feature_vector_from, const Features & to,
                const Features & from, std::unordered_map<std::size_t, std::size_t> & out_matches,
                typename FeatureVector::const_iterator to_first, typename FeatureVector::const_iterator to_last) const {
    const size_t nt = to.Size(), nf = from.Size();
    if (nt == 0) return;
    std::vector<int32_t> to_off{0}, from_off{0};
    std::vector<uint32_t> to_ids, from_ids;
    auto it_to = to_first;
    auto it_from = feature_vector_from.begin();
    while (it_to != to_last && it_to != feature_vector_to.end() && it_from != feature_vector_from.end()) {
      if (it_to->first == it_from->first) {
        to_ids.insert(to_ids.end(), it_to->second.begin(), it_to->second.end());
        from_ids.insert(from_ids.end(), it_from->second.begin(), it_from->second.end());
        to_off.push_back((int32_t)to_ids.size());
        from_off.push_back((int32_t)from_ids.size());
        ++it_to;
        ++it_from;
      } else if (it_to->first < it_from->first) {
        it_to = feature_vector_to.lower_bound(it_from->first);
      } else {
        it_from = feature_vector_from.lower_bound(it_to->first);
      }
    }
    orbgpu_group_args a{};
    a.to_desc = to.descriptors.data(); a.n_to = (int32_t)nt;
    a.from_desc = from.descriptors.data(); a.n_from = (int32_t)nf;
    a.n_groups = (int32_t)to_off.size() - 1;
    a.to_off = to_off.data(); a.to_ids = to_ids.data(); a.from_off = from_off.data(); a.from_ids = from_ids.data();
    a.threshold = THRESHOLD_; a.ratio = nearest_neighbour_ratio_;
    std::vector<int32_t> idx(nt);
    std::vector<uint32_t> dist(nt);
    if (orbgpu_match_groups(ctx_, &a, idx.data(), dist.data()) != ORBGPU_OK)
      throw std::runtime_error(std::string("orbgpu_match_groups: ") + orbgpu_last_error(ctx_));
    for (size_t i = 0; i < nt; ++i)
      if (idx[i] >= 0) out_matches[i] = (size_t)idx[i];
  }

 private:
  static void Fill(const Features & from, std::vector<double> & xy, std::vector<int32_t> & lev) {
    for (size_t j = 0; j < from.Size(); ++j) {
      xy[2 * j] = from.keypoints[j].X();      // distorted keypoints, features.cpp:104
      xy[2 * j + 1] = from.keypoints[j].Y();
      lev[j] = from.keypoints[j].level;
    }
  }
  template <class Emit>
  void Run(const uint8_t * to_desc, size_t nt, const std::vector<double> & to_xy, const std::vector<uint64_t> & win,
           const std::vector<int32_t> & lmin, const std::vector<int32_t> & lmax, const std::vector<uint8_t> * active,
           const Features & from, const std::vector<double> & from_xy, const std::vector<int32_t> & flev,
           Emit emit) const {
    if (nt == 0) return;
    orbgpu_window_args a{};
    a.to_desc = to_desc; a.to_xy = to_xy.data(); a.to_window = win.data();
    a.to_min_level = lmin.data(); a.to_max_level = lmax.data(); a.to_active = active ? active->data() : nullptr;
    a.n_to = (int32_t)nt;
    a.from_desc = from.descriptors.data(); a.from_xy = from_xy.data(); a.from_level = flev.data();
    a.n_from = (int32_t)from.Size();
    a.threshold = THRESHOLD_; a.ratio = nearest_neighbour_ratio_;
    std::vector<int32_t> idx(nt);
    std::vector<uint32_t> dist(nt);
    if (orbgpu_match_window(ctx_, &a, idx.data(), dist.data()) != ORBGPU_OK)
      throw std::runtime_error(std::string("orbgpu_match_window: ") + orbgpu_last_error(ctx_));
    for (size_t i = 0; i < nt; ++i)
      if (idx[i] >= 0) emit(i, (size_t)idx[i]);
  }
  orbgpu_ctx * ctx_;
  const precision_t nearest_neighbour_ratio_;
  const unsigned THRESHOLD_;
};

class HipOrientationValidator {
 public:
  HipOrientationValidator(orbgpu_ctx * ctx, const std::vector<KeyPoint> & kp_to, const std::vector<KeyPoint> & kp_from)
      : ctx_(ctx), kp_to_(&kp_to), kp_from_(&kp_from) {}
  // validators/orientation_validator.cpp:16-40: erases the matches outside the top-3 rotation bins, returns how many
  int Validate(std::unordered_map<std::size_t, std::size_t> & inout_matches_12) const {
    const size_t nt = kp_to_->size(), nf = kp_from_->size();
    if (nt == 0) return 0;
    std::vector<float> a_to(nt), a_from(nf);
    for (size_t i = 0; i < nt; ++i) a_to[i] = (float)(*kp_to_)[i].angle;      // angles hold float values
    for (size_t j = 0; j < nf; ++j) a_from[j] = (float)(*kp_from_)[j].angle;
    std::vector<int32_t> m(nt, -1);
    for (const auto & kv : inout_matches_12) m[kv.first] = (int32_t)kv.second;
    int32_t nd = 0;
    if (orbgpu_rotation_filter(ctx_, a_to.data(), (int32_t)nt, a_from.data(), (int32_t)nf, m.data(), &nd) != ORBGPU_OK)
      throw std::runtime_error(std::string("orbgpu_rotation_filter: ") + orbgpu_last_error(ctx_));
    for (auto it = inout_matches_12.begin(); it != inout_matches_12.end();)
      it = (m[it->first] < 0) ? inout_matches_12.erase(it) : std::next(it);
    return nd;
  }
 private:
  orbgpu_ctx * ctx_;
  const std::vector<KeyPoint> * kp_to_;
  const std::vector<KeyPoint> * kp_from_;
};

}  // namespace matching
}  // namespace features
}  // namespace orb_slam3
#endif
