// GPU-backed body for MapPoint::ComputeDistinctiveDescriptor (src/map/map_point.cpp:135-171), batched over the map
// points a caller refreshes together (LocalMapper after a keyframe insertion, Map::... loops that call it per point).
// Gathers each point's observation descriptors exactly as the reference does (Observation::AppendDescriptorsToList),
// makes ONE C-ABI call, and hands back the chosen descriptor per point.
#ifndef ORB_SLAM3_AMD_HIP_MAP_POINT_H_
#define ORB_SLAM3_AMD_HIP_MAP_POINT_H_

#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include <features/features.h>  // reference header: features::DescriptorType
#include "orbgpu.h"

namespace orb_slam3 {
namespace map {

// groups[p] = the descriptors of map point p in AppendDescriptorsToList order; result[p] = index of the distinctive
// descriptor inside groups[p], or -1 when the point has no observation (the reference then returns early, :146-148)
inline std::vector<int32_t> HipComputeDistinctiveDescriptors(orbgpu_ctx * ctx,
                                                             const std::vector<std::vector<features::DescriptorType>> & groups) {
  std::vector<int32_t> offsets(groups.size() + 1, 0);
  for (size_t p = 0; p < groups.size(); ++p) offsets[p + 1] = offsets[p] + (int32_t)groups[p].size();
  std::vector<uint8_t> desc((size_t)offsets.back() * 32);
  size_t row = 0;
  for (const auto & g : groups)
    for (const auto & d : g) std::memcpy(desc.data() + 32 * row++, d.data(), 32);
  std::vector<int32_t> best(groups.size(), -1);
  if (orbgpu_distinctive_descriptor(ctx, desc.data(), offsets.data(), (int32_t)groups.size(), best.data()) != ORBGPU_OK)
    throw std::runtime_error(std::string("orbgpu_distinctive_descriptor: ") + orbgpu_last_error(ctx));
  return best;
}

}  // namespace map
}  // namespace orb_slam3
#endif
