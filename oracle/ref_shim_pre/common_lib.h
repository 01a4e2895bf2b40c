// Build shim (OURS, not reference code).  Stands in for the reference's include/common_lib.h when the UNMODIFIED
// src/preprocess.cpp is compiled out-of-tree into oracle/_ref/libref_preprocess.so (oracle/Makefile `ref_pre`): only
// what that one translation unit uses — the point typedefs, LID_TYPE, the colour macros and a 3-vector with the handful
// of Eigen::Vector3d operations its (unused here) feature-extraction helpers call.  Eigen, PCL and ROS are not installed.
#pragma once
#include <omp.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include <pcl/point_types.h>
#include <pcl/point_cloud.h>

namespace Eigen {
struct Vector3d {
  double v[3];
  Vector3d() : v{0, 0, 0} {}
  Vector3d(double a, double b, double c) : v{a, b, c} {}
  static Vector3d Zero() { return Vector3d(); }
  double norm() const { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }
  double dot(const Vector3d& o) const { return v[0] * o.v[0] + v[1] * o.v[1] + v[2] * o.v[2]; }
  void setZero() { v[0] = v[1] = v[2] = 0; }
  struct RowT { const Vector3d* a; double operator*(const Vector3d& b) const { return a->dot(b); } };
  RowT transpose() const { return RowT{this}; }
  void normalize() { double n = norm(); v[0] /= n; v[1] /= n; v[2] /= n; }
  double& operator[](int i) { return v[i]; }
  double operator[](int i) const { return v[i]; }
  double& operator()(int i) { return v[i]; }
  Vector3d operator-(const Vector3d& o) const { return Vector3d(v[0] - o.v[0], v[1] - o.v[1], v[2] - o.v[2]); }
  Vector3d operator+(const Vector3d& o) const { return Vector3d(v[0] + o.v[0], v[1] + o.v[1], v[2] + o.v[2]); }
  Vector3d& operator-=(const Vector3d& o) { for (int i = 0; i < 3; i++) v[i] -= o.v[i]; return *this; }
  struct Comma {
    Vector3d* t; int k;
    Comma operator,(double x) { t->v[k] = x; return Comma{t, k + 1}; }
  };
  Comma operator<<(double x) { v[0] = x; return Comma{this, 1}; }
};
}  // namespace Eigen

using namespace std;
using namespace Eigen;

#define RESET "\033[0m"
#define BOLDRED "\033[1m\033[31m"

typedef pcl::PointXYZINormal PointType;
typedef pcl::PointCloud<PointType> PointCloudXYZI;

enum LID_TYPE { AVIA = 1, VELO, OUSTER, L515, PANDAR, ROBOSENSE };  // same values as the reference header (common_lib.h:55)
