// Build shim (OURS, not reference code, not PCL): the two point types include/common_lib.h typedefs, default-initialised (the reference
// ikd-Tree declares `const PointType ZeroP;`), layout of pcl::PointXYZINormal (48 bytes, 16-aligned) - and PCL's DEG2RAD macro
// (pcl/common/angles.h), which the reference's calcBodyVar (src/laserMapping.cpp:163) uses.  Test infrastructure only.
#pragma once
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <deque>
#include <string>
#include <vector>
namespace pcl {
struct alignas(16) PointXYZINormal {
  float x = 0.f, y = 0.f, z = 0.f, _p0 = 1.f;
  float normal_x = 0.f, normal_y = 0.f, normal_z = 0.f, _p1 = 0.f;
  float intensity = 0.f, curvature = 0.f, _p2 = 0.f, _p3 = 0.f;
};
struct alignas(16) PointXYZRGB { float x = 0.f, y = 0.f, z = 0.f, _p0 = 1.f; std::uint32_t rgba = 0; float _p1 = 0.f, _p2 = 0.f, _p3 = 0.f; };
static_assert(sizeof(PointXYZINormal) == 48, "layout");
}  // namespace pcl
#ifndef DEG2RAD
#define DEG2RAD(x) ((x) * 0.017453293)
#endif
