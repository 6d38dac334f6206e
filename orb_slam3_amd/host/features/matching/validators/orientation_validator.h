// REPLACES src/features/matching/validators/orientation_validator.h (with orientation_validator.cpp: drop it from
// src/CMakeLists.txt) in the reference tree.  Same class, same two-argument constructor, same Validate -- the callers
// (frame/monocular/monocular_frame.cpp:186-189, features/handlers/DBoW2/dbo_w2_handler.cpp:68-71) compile unchanged.
// The rotation histogram (orientation_validator.cpp:16-97) runs on the GPU through orbgpu_rotation_filter, on the
// context of the process's HipORBFeatureExtractor (the reference has exactly one extractor, owned by the Atlas).
#ifndef ORB_SLAM3_ORB_SLAM3_INCLUDE_FEATURES_MATCHING_VALIDATORS_ORIENTATION_VALIDATOR_H_
#define ORB_SLAM3_ORB_SLAM3_INCLUDE_FEATURES_MATCHING_VALIDATORS_ORIENTATION_VALIDATOR_H_

#include <stdexcept>
#include <unordered_map>

#include <features/features.h>
#include "hip_orb_feature_extractor.h"
#include "hip_snn_matcher.h"

namespace orb_slam3 {
namespace features {
namespace matching {

class OrientationValidator {
 public:
  OrientationValidator(const std::vector<KeyPoint> & kp_to, const std::vector<KeyPoint> & kp_from)
      : kp_to_(&kp_to), kp_from_(&kp_from) {}
  // erases the matches outside the three dominant rotation bins; returns how many it erased (:16-40)
  int Validate(std::unordered_map<std::size_t, std::size_t> & inout_matches_12) {
    orbgpu_ctx * ctx = HipORBFeatureExtractor::DefaultContext();
    if (!ctx) throw std::runtime_error("OrientationValidator: no HipORBFeatureExtractor exists (there is no CPU path)");
    return HipOrientationValidator(ctx, *kp_to_, *kp_from_).Validate(inout_matches_12);
  }

 private:
  const std::vector<KeyPoint> * kp_to_;
  const std::vector<KeyPoint> * kp_from_;
};

}  // namespace matching
}  // namespace features
}  // namespace orb_slam3
#endif
