// REPLACES src/features/matching/second_nearest_neighbor_matcher.hpp in the reference tree.
//
// Same class template, same constructor, same MatchWithIterators(to_begin, to_end, feature_extractor, out_matches)
// and the same MatchMapType / FromIdType / ToIdType typedefs, so that the three callers --
//   MonocularFrame::ComputeMatchesForLinking            frame/monocular/monocular_frame.cpp:173-184  (AreaToIterator)
//   MonocularFrame::SearchInVisiblePoints                monocular_frame.cpp:285-306                 (ProjectionSearchIterator)
//   MonocularKeyFrame::MatchVisibleMapPoints / AdjustSim3Transformation   monocular_key_frame.cpp:183-216, 353-376
//   DBoW2Handler::FastMatch                              features/handlers/DBoW2/dbo_w2_handler.cpp:25-72 (BowToIterator)
// -- compile and behave unchanged.  The body of the reference (:30-80: per 'to' a scan of its candidates, best / second
// best, ratio test, per-'from' uniqueness) exists once per iterator type as an explicit specialisation that reads the
// range out of the two iterators, takes the GPU context from the HipORBFeatureExtractor behind `feature_extractor`
// and makes ONE C-ABI call (orbgpu_match_window / orbgpu_match_groups).  There is no CPU body: an iterator type
// without a specialisation does not compile, an extractor that is not a HipORBFeatureExtractor throws.
#ifndef ORB_SLAM3_WINDOW_MATCHER_H
#define ORB_SLAM3_WINDOW_MATCHER_H

#include <stdexcept>
#include <unordered_map>

#include <features/ifeature_extractor.h>
#include <features/matching/iterators/area_to_iterator.h>
#include <features/matching/iterators/projection_search_iterator.h>
#include <features/handlers/DBoW2/bow_to_iterator.h>
#include "hip_orb_feature_extractor.h"
#include "hip_snn_matcher.h"

namespace orb_slam3 {
namespace features {
namespace matching {

namespace detail {
template <class T>
struct no_gpu_body { static constexpr bool value = false; };
inline orbgpu_ctx * context_of(const IFeatureExtractor * feature_extractor) {
  const auto * hip = dynamic_cast<const HipORBFeatureExtractor *>(feature_extractor);
  if (!hip) throw std::runtime_error("SNNMatcher: the feature extractor is not a HipORBFeatureExtractor (there is no CPU path)");
  return hip->context();
}
}  // namespace detail

template <typename IteratorType>
class SNNMatcher {
 public:
  SNNMatcher(const precision_t, const unsigned) {}
  template <class Map>
  void MatchWithIterators(IteratorType, IteratorType, const IFeatureExtractor *, Map &) {
    static_assert(detail::no_gpu_body<IteratorType>::value,
                  "SNNMatcher has GPU bodies for AreaToIterator, ProjectionSearchIterator and BowToIterator only");
  }
};

// ---- window search, level 0 on both sides (area_to_iterator.h:16-58, area_to_pointee.h:21-62) ----------------------
template <>
class SNNMatcher<iterators::AreaToIterator> {
 public:
  typedef std::size_t ToIdType;
  typedef std::size_t FromIdType;
  typedef std::unordered_map<ToIdType, FromIdType> MatchMapType;
  SNNMatcher(const precision_t nearest_neighbour_ratio, const unsigned threshold)
      : nearest_neighbour_ratio_(nearest_neighbour_ratio), THRESHOLD_(threshold) {}
  void MatchWithIterators(iterators::AreaToIterator to_begin, iterators::AreaToIterator to_end,
                          const IFeatureExtractor * feature_extractor, MatchMapType & out_matches) {
    HipSNNMatcher(detail::context_of(feature_extractor), nearest_neighbour_ratio_, THRESHOLD_)
        .MatchArea(*to_begin.FeaturesTo(), *to_begin.FeaturesFrom(), to_begin.WindowSize(), out_matches, to_begin.GetId(),
                   to_end.GetId());
  }

 private:
  const precision_t nearest_neighbour_ratio_;
  const unsigned THRESHOLD_;
};

// ---- search by projection (projection_search_iterator.cpp:12-27, projection_search_pointee.cpp:15-35) --------------
template <>
class SNNMatcher<iterators::ProjectionSearchIterator> {
 public:
  typedef map::MapPoint * ToIdType;
  typedef std::size_t FromIdType;
  typedef std::unordered_map<ToIdType, FromIdType> MatchMapType;
  SNNMatcher(const precision_t nearest_neighbour_ratio, const unsigned threshold)
      : nearest_neighbour_ratio_(nearest_neighbour_ratio), THRESHOLD_(threshold) {}
  void MatchWithIterators(iterators::ProjectionSearchIterator to_begin, iterators::ProjectionSearchIterator to_end,
                          const IFeatureExtractor * feature_extractor, MatchMapType & out_matches) {
    HipSNNMatcher(detail::context_of(feature_extractor), nearest_neighbour_ratio_, THRESHOLD_)
        .MatchProjection(to_begin.Position(), to_end.Position(), *to_begin.FeaturesFrom(), out_matches);
  }

 private:
  const precision_t nearest_neighbour_ratio_;
  const unsigned THRESHOLD_;
};

// ---- candidates of the same vocabulary node (bow_to_iterator.{h,cpp}, bow_to_pointee.h:19-59) ----------------------
template <>
class SNNMatcher<handlers::iterators::BowToIterator> {
 public:
  typedef std::size_t ToIdType;
  typedef std::size_t FromIdType;
  typedef std::unordered_map<ToIdType, FromIdType> MatchMapType;
  SNNMatcher(const precision_t nearest_neighbour_ratio, const unsigned threshold)
      : nearest_neighbour_ratio_(nearest_neighbour_ratio), THRESHOLD_(threshold) {}
  void MatchWithIterators(handlers::iterators::BowToIterator to_begin, handlers::iterators::BowToIterator to_end,
                          const IFeatureExtractor * feature_extractor, MatchMapType & out_matches) {
    HipSNNMatcher(detail::context_of(feature_extractor), nearest_neighbour_ratio_, THRESHOLD_)
        .MatchBow(*to_begin.FeatureVectorTo(), *to_begin.FeatureVectorFrom(), *to_begin.FeaturesTo(), *to_begin.FeaturesFrom(),
                  out_matches, to_begin.Position(), to_end.Position());
  }

 private:
  const precision_t nearest_neighbour_ratio_;
  const unsigned THRESHOLD_;
};

}  // namespace matching
}  // namespace features
}  // namespace orb_slam3
#endif  // ORB_SLAM3_WINDOW_MATCHER_H
