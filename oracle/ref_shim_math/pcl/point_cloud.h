// Build shim (OURS): the part of pcl::PointCloud<T> that src/preprocess.cpp touches.
#pragma once
#include <cstdint>
#include <memory>
#include <vector>

namespace pcl {
template <class T>
struct PointCloud {
  typedef std::shared_ptr<PointCloud<T>> Ptr;
  std::vector<T> points;
  std::uint32_t width = 0, height = 0;
  bool is_dense = true;
  PointCloud() {}
  PointCloud(std::uint32_t w, std::uint32_t h) : points(std::size_t(w) * h), width(w), height(h) {}  // (src/laserMapping.cpp:117-119)
  bool empty() const { return points.empty(); }
  std::size_t size() const { return points.size(); }
  void clear() { points.clear(); width = height = 0; }
  void reserve(std::size_t n) { points.reserve(n); }
  void resize(std::size_t n) { points.resize(n); }
  void push_back(const T& p) { points.push_back(p); }
  T& operator[](std::size_t i) { return points[i]; }
  const T& operator[](std::size_t i) const { return points[i]; }
  typename std::vector<T>::iterator begin() { return points.begin(); }
  typename std::vector<T>::iterator end() { return points.end(); }
  PointCloud& operator+=(const PointCloud& o) { points.insert(points.end(), o.points.begin(), o.points.end()); return *this; }
};
}  // namespace pcl
