// REPLACES src/features/matching/iterators/projection_search_iterator.h (with projection_search_iterator.cpp,
// projection_search_pointee.{h,cpp}, which it no longer needs: drop the two .cpp files from src/CMakeLists.txt) in
// the reference tree.  Same class name and constructor -- the callers (frame/monocular/monocular_frame.cpp:285-306,
// monocular_key_frame.cpp:183-216, 353-376) compile unchanged -- but the iterator only carries the list range and the
// 'from' features; SNNMatcher<ProjectionSearchIterator> evaluates the range in one call on the GPU.
#ifndef ORB_SLAM3_ORB_SLAM3_INCLUDE_FEATURES_MATCHING_ITERATORS_PROJECTION_SEARCH_ITERATOR_H_
#define ORB_SLAM3_ORB_SLAM3_INCLUDE_FEATURES_MATCHING_ITERATORS_PROJECTION_SEARCH_ITERATOR_H_

#include <list>

#include <features/features.h>
#include <frame/map_point_visibility_params.h>
#include <map/map_point.h>

namespace orb_slam3 {
namespace features {
namespace matching {
namespace iterators {

class ProjectionSearchIterator {
 public:
  typedef map::MapPoint * id_type;  // ProjectionSearchPointee::id_type, projection_search_pointee.h:24
  typedef std::list<frame::MapPointVisibilityParams>::const_iterator list_iterator;
  ProjectionSearchIterator(list_iterator begin, list_iterator end, const Features * from_features)
      : it_(begin), end_(end), from_features_(from_features) {}

  list_iterator Position() const { return it_; }
  const Features * FeaturesFrom() const { return from_features_; }

  friend bool operator==(const ProjectionSearchIterator & a, const ProjectionSearchIterator & b) { return a.it_ == b.it_; }
  friend bool operator!=(const ProjectionSearchIterator & a, const ProjectionSearchIterator & b) { return a.it_ != b.it_; }

 private:
  list_iterator it_, end_;
  const Features * from_features_;
};

}  // namespace iterators
}  // namespace matching
}  // namespace features
}  // namespace orb_slam3
#endif
