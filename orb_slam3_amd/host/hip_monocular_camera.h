// GPU-backed body for the per-keypoint loop of FeatureHandlerFactory::Create
// (src/factories/feature_handler_factory.cpp:53-60): MonocularCamera::UndistortPoint and ::UnprojectAndUndistort
// (src/camera/monocular_camera.cpp:40-61) for every keypoint of a Features object, in ONE C-ABI call.
// The camera is passed as plain numbers (orbgpu_camera); INTEGRATION.md shows how to fill it from
// camera::MonocularCamera and its IDistortionModel.
#ifndef ORB_SLAM3_AMD_HIP_MONOCULAR_CAMERA_H_
#define ORB_SLAM3_AMD_HIP_MONOCULAR_CAMERA_H_

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include <features/features.h>  // reference header
#include "orbgpu.h"

namespace orb_slam3 {
namespace camera {

// fills features.undistorted_keypoints / undistorted_and_unprojected_keypoints (resized like :54-55); returns, per
// keypoint, bit 0 / bit 1 = what UndistortPoint / UnprojectAndUndistort return (the reference drops both bools)
inline std::vector<uint8_t> HipUndistortFeatures(orbgpu_ctx * ctx, const orbgpu_camera & cam, features::Features & features) {
  const size_t n = features.Size();
  features.undistorted_keypoints.resize(n);
  features.undistorted_and_unprojected_keypoints.resize(n);
  std::vector<uint8_t> ok(n);
  if (n == 0) return ok;
  std::vector<orbgpu_keypoint> kp(n);
  for (size_t i = 0; i < n; ++i) {
    kp[i].x = features.keypoints[i].pt[0];
    kp[i].y = features.keypoints[i].pt[1];
    kp[i].angle = 0; kp[i].size = 0; kp[i].level = features.keypoints[i].level; kp[i].reserved = 0;
  }
  std::vector<double> und(2 * n), unp(3 * n);
  if (orbgpu_keypoint_geometry(ctx, &cam, kp.data(), (int32_t)n, und.data(), unp.data(), ok.data()) != ORBGPU_OK)
    throw std::runtime_error(std::string("orbgpu_keypoint_geometry: ") + orbgpu_last_error(ctx));
  for (size_t i = 0; i < n; ++i) {
    features.undistorted_keypoints[i][0] = und[2 * i];
    features.undistorted_keypoints[i][1] = und[2 * i + 1];
    for (int c = 0; c < 3; ++c) features.undistorted_and_unprojected_keypoints[i][c] = unp[3 * i + c];
  }
  return ok;
}

}  // namespace camera
}  // namespace orb_slam3
#endif
