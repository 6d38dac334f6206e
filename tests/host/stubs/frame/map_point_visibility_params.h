// TEST STAND-IN for src/frame/map_point_visibility_params.h:16-22 (same fields).
#pragma once
#include <features/features.h>
namespace orb_slam3 { namespace map { class MapPoint; } namespace frame {
struct MapPointVisibilityParams {
  map::MapPoint * map_point;
  TPoint3D mp_world_pos;
  precision_t window_size;
  TPoint2D position;
  int level;
};
}}
