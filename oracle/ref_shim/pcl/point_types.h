// Build shim (OURS, not reference code): the minimum of <pcl/point_types.h> that the reference's
// include/ikd-Tree/ikd_Tree.h needs, so the UNMODIFIED reference tree can be compiled out-of-tree
// into oracle/_ref/ as a cross-check for the oracle.  Layout = pcl::PointXYZINormal:
// 48 bytes, 16-byte aligned: {x,y,z,pad} {normal_x,normal_y,normal_z,pad} {intensity,curvature,pad,pad}.
#pragma once
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

namespace pcl {
struct alignas(16) PointXYZINormal {
  float x = 0.f, y = 0.f, z = 0.f, data_pad = 1.f;
  float normal_x = 0.f, normal_y = 0.f, normal_z = 0.f, normal_pad = 0.f;
  float intensity = 0.f, curvature = 0.f, tail_pad0 = 0.f, tail_pad1 = 0.f;
};
static_assert(sizeof(PointXYZINormal) == 48, "layout");
}  // namespace pcl
