// Build shim (OURS): the two point types common_lib.h typedefs (layout of pcl::PointXYZINormal: 48 bytes, 16-aligned).
#pragma once
#include <cstdint>
#include <deque>
#include <string>
#include <vector>
namespace pcl {
struct alignas(16) PointXYZINormal {
  float x, y, z, _p0;
  float normal_x, normal_y, normal_z, _p1;
  float intensity, curvature, _p2, _p3;
};
struct alignas(16) PointXYZRGB { float x, y, z, _p0; std::uint32_t rgba; float _p1, _p2, _p3; };
}  // namespace pcl
