// TEST STAND-IN for src/map/map_point.h: only GetDescriptor() is used by the matcher adapter.
#pragma once
#include <features/features.h>
namespace orb_slam3 { namespace map {
class MapPoint {
 public:
  features::DescriptorType desc;
  const features::DescriptorType GetDescriptor() const { return desc; }
};
}}
