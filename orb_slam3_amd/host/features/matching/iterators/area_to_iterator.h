// REPLACES src/features/matching/iterators/area_to_iterator.h (and area_to_pointee.h, which it no longer needs) in the
// reference tree.  Same class name, same constructor, same comparison operators -- the caller
// (frame/monocular/monocular_frame.cpp:175-182) compiles unchanged -- but the iterator only CARRIES the arguments of
// the search: SNNMatcher<AreaToIterator> (second_nearest_neighbor_matcher.hpp, this directory) evaluates the whole
// range [begin, end) in one call on the GPU, so nothing is scanned on the CPU per step and the reference's
// out-of-bounds read of keypoints[Size()] in the end iterator's constructor (area_to_iterator.h:24-28,
// area_to_pointee.h:44-48) does not happen.
#ifndef ORB_SLAM3_ORB_SLAM3_INCLUDE_FEATURES_MATCHING_ITERATORS_AREA_TO_ITERATOR_H_
#define ORB_SLAM3_ORB_SLAM3_INCLUDE_FEATURES_MATCHING_ITERATORS_AREA_TO_ITERATOR_H_

#include <algorithm>
#include <cstddef>

#include <features/features.h>

namespace orb_slam3 {
namespace features {
namespace matching {
namespace iterators {

class AreaToIterator {
 public:
  typedef size_t id_type;
  AreaToIterator() = default;
  AreaToIterator(id_type id, const Features * features_to, const Features * features_from, size_t window_size)
      : id_(std::min(features_to->Size(), id)),  // area_to_iterator.h:27
        features_to_(features_to), features_from_(features_from), window_size_(window_size) {}

  // what the matcher reads (stays inside src/features)
  id_type GetId() const { return id_; }
  const Features * FeaturesTo() const { return features_to_; }
  const Features * FeaturesFrom() const { return features_from_; }
  size_t WindowSize() const { return window_size_; }

  friend bool operator==(const AreaToIterator & a, const AreaToIterator & b) { return a.id_ == b.id_; }
  friend bool operator!=(const AreaToIterator & a, const AreaToIterator & b) { return a.id_ != b.id_; }

 private:
  id_type id_ = 0;
  const Features * features_to_ = nullptr;
  const Features * features_from_ = nullptr;
  size_t window_size_ = 0;
};

}  // namespace iterators
}  // namespace matching
}  // namespace features
}  // namespace orb_slam3
#endif
