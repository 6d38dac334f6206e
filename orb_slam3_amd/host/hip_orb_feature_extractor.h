// HipORBFeatureExtractor -- drop-in for features::ORBFeatureExtractor over the orbgpu C ABI.
//
// Compiled INSIDE the reference tree (it includes the reference's own headers): add this file and
// include/orbgpu.h to src/features/, link liborbgpu.so, and construct it wherever the reference constructs
// ORBFeatureExtractor (main.cpp:451-454, factories/feature_extractor_factory.cpp:10-17).  Frame / KeyFrame
// construction, Tracker, LocalMapper and LoopMergeDetector stay unchanged: they only see IFeatureExtractor
// (src/features/ifeature_extractor.h:24-60).
//
// Every virtual keeps the reference's meaning (orb_feature_extractor.cpp line cited per method).
#ifndef ORB_SLAM3_AMD_HIP_ORB_FEATURE_EXTRACTOR_H_
#define ORB_SLAM3_AMD_HIP_ORB_FEATURE_EXTRACTOR_H_

#include <cmath>
#include <mutex>
#include <stdexcept>
#include <vector>

#include <features/ifeature_extractor.h>  // reference header
#include "orbgpu.h"

namespace orb_slam3 {
namespace features {

class HipORBFeatureExtractor : public IFeatureExtractor {
 public:
  // same leading arguments as ORBFeatureExtractor (orb_feature_extractor.h:24-29)
  HipORBFeatureExtractor(unsigned image_width, unsigned image_height, precision_t scale_factor, size_t levels,
                         unsigned init_threshold_FAST, unsigned min_threshold_FAST, int device = 0,
                         int max_features = 8000)
      : image_width_(image_width), image_height_(image_height), scale_factor_(scale_factor),
        log_scale_factor_(std::log(scale_factor)), init_threshold_FAST_(init_threshold_FAST),
        min_threshold_FAST_(min_threshold_FAST), scale_factors_(levels), level_sigma2_(levels) {
    orbgpu_params p{};
    p.width = (int32_t)image_width; p.height = (int32_t)image_height; p.scale_factor = scale_factor;
    p.levels = (int32_t)levels; p.fast_init = (int32_t)init_threshold_FAST; p.fast_min = (int32_t)min_threshold_FAST;
    p.device = device; p.max_batch = 1; p.max_features = max_features;
    if (orbgpu_create(&p, &ctx_) != ORBGPU_OK)  // the code base throws std::runtime_error elsewhere
      throw std::runtime_error(std::string("orbgpu_create: ") + orbgpu_last_error(nullptr));
    orbgpu_geometry g;
    orbgpu_get_geometry(ctx_, max_features, &g);
    for (size_t i = 0; i < levels; ++i) {  // Initialize(), orb_feature_extractor.cpp:53-64
      scale_factors_[i] = g.scale[i];
      level_sigma2_[i] = g.sigma2[i];
    }
    max_features_ = max_features;
    if (!DefaultContextSlot()) DefaultContextSlot() = ctx_;
  }
  ~HipORBFeatureExtractor() override {
    if (DefaultContextSlot() == ctx_) DefaultContextSlot() = nullptr;
    orbgpu_destroy(ctx_);
  }
  HipORBFeatureExtractor(const HipORBFeatureExtractor &) = delete;
  HipORBFeatureExtractor & operator=(const HipORBFeatureExtractor &) = delete;

  // orb_feature_extractor.cpp:530-600.  Returns -1 on an empty image, otherwise monoIndex (= 0, every
  // keypoint takes the reversed "stereo" branch, :550,579-587); the caller ignores it
  // (factories/feature_handler_factory.cpp:50).
  int Extract(const TImageGray8U & image, Features & out_features, size_t feature_count) const override {
    if (image.rows() == 0 || image.cols() == 0) return -1;
    orbgpu_geometry g;
    if (orbgpu_get_geometry(ctx_, (int32_t)feature_count, &g) != ORBGPU_OK) return -1;
    std::vector<orbgpu_keypoint> kp((size_t)g.keypoint_capacity);
    std::vector<uint8_t> desc((size_t)g.keypoint_capacity * 32);
    int32_t n = 0;
    int rc = orbgpu_extract(ctx_, image.data(), (int32_t)image.cols(), (int32_t)image.rows(), (int32_t)image.cols(),
                            (int32_t)feature_count, kp.data(), desc.data(), g.keypoint_capacity, &n);
    if (rc == ORBGPU_E_EMPTY) return -1;
    if (rc != ORBGPU_OK) throw std::runtime_error(std::string("orbgpu_extract: ") + orbgpu_last_error(ctx_));
    out_features.descriptors.resize(n, 32);  // :545-546
    out_features.keypoints.resize((size_t)n);
    for (int32_t i = 0; i < n; ++i) {
      KeyPoint & k = out_features.keypoints[(size_t)i];
      k.pt[0] = kp[(size_t)i].x;
      k.pt[1] = kp[(size_t)i].y;
      k.angle = (precision_t)kp[(size_t)i].angle;  // float widened to double like :201
      k.size = (precision_t)kp[(size_t)i].size;
      k.level = kp[(size_t)i].level;
    }
    if (n) std::memcpy(out_features.descriptors.data(), desc.data(), (size_t)n * 32);  // RowMajor N x 32
    return 0;
  }

  FeatureExtractorType Type() const override { return ORBFE; }  // :71-73

  precision_t GetAcceptableSquareError(unsigned level) const override {  // :654-660
    if (level >= level_sigma2_.size()) return 0;
    return level_sigma2_[level];
  }

  void ComputeInvariantDistances(const TPoint3D & point, const KeyPoint & key_point, precision_t & out_max_distance,
                                 precision_t & out_min_distance) const override {  // :854-862
    precision_t distance = point.norm();
    out_max_distance = distance * scale_factors_[(size_t)key_point.level];
    out_min_distance = out_max_distance / scale_factors_.back();
  }

  unsigned PredictScale(precision_t distance, precision_t max_distance) const override {  // :864-873
    precision_t ratio = max_distance / distance;
    unsigned scale = (unsigned)std::max(0., std::ceil(std::log(ratio) / log_scale_factor_));
    if (scale >= scale_factors_.size()) scale = (unsigned)scale_factors_.size() - 1;
    return scale;
  }

  // :83-98 -- called from SNNMatcher and MapPoint on three threads; pure host popcount, no GPU call
  unsigned ComputeDistance(const DescriptorType & d1, const DescriptorType & d2) const override {
    return orbgpu_hamming256(d1.data(), d2.data());
  }

  const std::vector<precision_t> & GetScaleFactors() const override { return scale_factors_; }
  precision_t GetHighThreshold() const override { return 100; }  // :75-77
  precision_t GetLowThreshold() const override { return 50; }    // :79-81

  void Serialize(std::ostream & ostream) const override {  // :875-885, readable by ORBFeatureExtractor(istream&)
    WRITE_TO_STREAM(image_width_, ostream);
    WRITE_TO_STREAM(image_height_, ostream);
    WRITE_TO_STREAM(scale_factor_, ostream);
    size_t levels = scale_factors_.size();
    WRITE_TO_STREAM(levels, ostream);
    WRITE_TO_STREAM(init_threshold_FAST_, ostream);
    WRITE_TO_STREAM(min_threshold_FAST_, ostream);
  }

  orbgpu_ctx * context() const { return ctx_; }
  // The context of the process's first live extractor: what callers without an extractor argument use
  // (features::matching::OrientationValidator has the reference's two-argument constructor).  The reference has one
  // extractor per process, owned by the Atlas (map/atlas.h:44).
  static orbgpu_ctx * DefaultContext() { return DefaultContextSlot(); }

 private:
  static orbgpu_ctx *& DefaultContextSlot() {
    static orbgpu_ctx * slot = nullptr;
    return slot;
  }
  unsigned image_width_, image_height_;
  precision_t scale_factor_, log_scale_factor_;
  unsigned init_threshold_FAST_, min_threshold_FAST_;
  std::vector<precision_t> scale_factors_, level_sigma2_;
  int max_features_ = 0;
  orbgpu_ctx * ctx_ = nullptr;
};

}  // namespace features
}  // namespace orb_slam3
#endif
