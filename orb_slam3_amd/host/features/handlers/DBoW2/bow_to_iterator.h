// REPLACES src/features/handlers/DBoW2/bow_to_iterator.h (with bow_to_iterator.cpp and bow_to_pointee.h, which it no
// longer needs: drop the .cpp from src/CMakeLists.txt) in the reference tree.  Same class name and constructor -- the
// caller (features/handlers/DBoW2/dbo_w2_handler.cpp:53-65) compiles unchanged -- but the iterator only carries the two
// feature vectors and feature sets; SNNMatcher<BowToIterator> joins the vectors (what AdvanceUntilSameNode did,
// bow_to_iterator.cpp:27-47) and evaluates every shared node in one call on the GPU.
#ifndef ORB_SLAM3_ORB_SLAM3_INCLUDE_FEATURES_MATCHING_ITERATORS_BOW_TO_ITERATOR_H_
#define ORB_SLAM3_ORB_SLAM3_INCLUDE_FEATURES_MATCHING_ITERATORS_BOW_TO_ITERATOR_H_

#include <DBoW2/FeatureVector.h>

#include <features/features.h>

namespace orb_slam3 {
namespace features {
namespace handlers {
namespace iterators {

class BowToIterator {
 public:
  typedef size_t id_type;  // BowToPointee::id_type, bow_to_pointee.h:21
  BowToIterator(DBoW2::FeatureVector::const_iterator begin, const DBoW2::FeatureVector * feature_vector_to,
                const DBoW2::FeatureVector * feature_vector_from, const features::Features * features_to,
                const features::Features * features_from)
      : bow_it_to_(begin), feature_vector_to_(feature_vector_to), feature_vector_from_(feature_vector_from),
        features_to_(features_to), features_from_(features_from) {}

  DBoW2::FeatureVector::const_iterator Position() const { return bow_it_to_; }
  const DBoW2::FeatureVector * FeatureVectorTo() const { return feature_vector_to_; }
  const DBoW2::FeatureVector * FeatureVectorFrom() const { return feature_vector_from_; }
  const features::Features * FeaturesTo() const { return features_to_; }
  const features::Features * FeaturesFrom() const { return features_from_; }

  friend bool operator==(const BowToIterator & a, const BowToIterator & b) { return a.bow_it_to_ == b.bow_it_to_; }
  friend bool operator!=(const BowToIterator & a, const BowToIterator & b) { return !(a == b); }

 private:
  DBoW2::FeatureVector::const_iterator bow_it_to_;
  const DBoW2::FeatureVector * feature_vector_to_;
  const DBoW2::FeatureVector * feature_vector_from_;
  const features::Features * features_to_;
  const features::Features * features_from_;
};

}  // namespace iterators
}  // namespace handlers
}  // namespace features
}  // namespace orb_slam3
#endif
