// TEST STAND-IN for the reference's src/features headers (typedefs.h, key_point.h, features.h,
// feature_extractor_type.h, ifeature_extractor.h): just enough surface (rows/cols/data/resize, KeyPoint, the
// IFeatureExtractor virtuals with the reference's signatures) to compile the adapters of orb_slam3_amd/host/ in
// this image, where Eigen and the rest of the reference's dependencies do not exist.  Not part of the product.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <istream>
#include <ostream>
#include <vector>

namespace orb_slam3 {
#define WRITE_TO_STREAM(num, stream) stream.write((char *)(&num), sizeof(num));
typedef double precision_t;

template <class T>
struct RowMajorMatrix {
  std::vector<T> v;
  long r = 0, c = 0;
  long rows() const { return r; }
  long cols() const { return c; }
  T * data() { return v.data(); }
  const T * data() const { return v.data(); }
  void resize(long rr, long cc) { r = rr; c = cc; v.resize((size_t)(rr * cc)); }
};
typedef RowMajorMatrix<uint8_t> TImageGray8U;

template <int N>
struct Vec {
  double d[N] = {};
  double & operator[](int i) { return d[i]; }
  const double & operator[](int i) const { return d[i]; }
  const double & x() const { return d[0]; }
  const double & y() const { return d[1]; }
  double norm() const { double s = 0; for (int i = 0; i < N; ++i) s += d[i] * d[i]; return std::sqrt(s); }
};
typedef Vec<2> TPoint2D;
typedef Vec<3> TPoint3D;

namespace serialization { class SerializationContext; }

namespace features {
enum FeatureExtractorType { ORBFE };

struct KeyPoint {  // src/features/key_point.h:10-23
  KeyPoint() = default;
  const precision_t & X() const { return pt.x(); }
  const precision_t & Y() const { return pt.y(); }
  TPoint2D pt;
  precision_t angle = 0;
  precision_t size = 0;
  int level = 0;
};

typedef RowMajorMatrix<uint8_t> DescriptorSet;
struct DescriptorType {  // Eigen::Matrix<uint8_t, 1, 32, RowMajor>
  uint8_t d[32];
  const uint8_t * data() const { return d; }
  uint8_t * data() { return d; }
};

class Features {  // src/features/features.h:23-54
 public:
  Features(precision_t width, precision_t height) : image_width(width), image_height(height) {}
  DescriptorSet descriptors;
  std::vector<KeyPoint> keypoints;
  std::vector<TPoint2D> undistorted_keypoints;
  std::vector<TPoint3D> undistorted_and_unprojected_keypoints;
  precision_t image_width, image_height;
  size_t Size() const { return keypoints.size(); }
};

class IFeatureExtractor {  // src/features/ifeature_extractor.h:24-60
 public:
  virtual int Extract(const TImageGray8U & image, Features & out_features, size_t feature_count) const = 0;
  virtual FeatureExtractorType Type() const = 0;
  virtual precision_t GetAcceptableSquareError(unsigned level) const = 0;
  virtual void ComputeInvariantDistances(const TPoint3D & point, const KeyPoint & key_point,
                                         precision_t & out_max_distance, precision_t & out_min_distance) const = 0;
  virtual unsigned PredictScale(precision_t distance, precision_t max_distance) const = 0;
  virtual unsigned ComputeDistance(const DescriptorType & d1, const DescriptorType & d2) const = 0;
  virtual const std::vector<precision_t> & GetScaleFactors() const = 0;
  virtual precision_t GetHighThreshold() const = 0;
  virtual precision_t GetLowThreshold() const = 0;
  virtual void Serialize(std::ostream & ostream) const = 0;
  virtual ~IFeatureExtractor() = default;
};
}  // namespace features
}  // namespace orb_slam3
