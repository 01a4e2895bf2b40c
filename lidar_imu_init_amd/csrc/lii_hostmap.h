// Host-side local map of libliinit_hip: the SET semantics of the reference's incremental k-d tree
// (build, add with per-voxel "keep the point closest to the voxel centre" down-sampling, add without)
// re-designed as a voxel hash — the device answers the nearest-neighbour queries, so the host only needs
// the box query that Add_Points performs.
//   Add_Points ........ reference include/ikd-Tree/ikd_Tree.cpp:381-456
//   box predicate ..... Search_by_range / Delete_by_range, ikd_Tree.cpp:616-629, :970-985
//                       (vertex_min <= p && vertex_max > p per axis, float arithmetic)
//   same_point ........ ikd_Tree.cpp:1269-1271 (EPSS 1e-6)
// Exactness: the reference's voxel box is [fl(i*ds), fl(fl(i*ds)+ds)) with i = floor(fl(p/ds)) in float32.
// A stored point is filed under its nominal voxel when it lies in that box and in neither neighbour box on
// every axis ("regular"); the (measure ~1e-6) remainder goes to a small `ambiguous_` list that every box
// query also scans with the exact predicate.  The resulting point set is identical to the tree's.
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

namespace lii {

class HostVoxelMap {
 public:
  void set_downsample(float ds) { ds_ = ds; }
  float downsample() const { return ds_; }
  int valid() const { return n_alive_; }
  uint64_t version() const { return version_; }

  void clear() {
    xyz_.clear(); next_.clear(); key_.clear(); state_.clear(); free_.clear(); ambiguous_.clear();
    table_.assign(1024, -1);
    tkeys_.assign(1024, 0);
    n_buckets_ = 0;
    n_alive_ = 0;
    version_++;
  }

  // Build(point_cloud): plain insertion of every point (no down-sampling) — ikd_Tree.cpp:336-347
  void build(const float* xyz, int n, int stride_floats) {
    clear();
    for (int i = 0; i < n; i++) insert(xyz + size_t(i) * stride_floats);
  }

  // Add_Points(PointToAdd, downsample_on); returns the reference's tmp_counter
  int add_points(const float* xyz, int n, int stride_floats, bool downsample_on) {
    int counter = 0;
    std::vector<int> in_box;
    for (int i = 0; i < n; i++) {
      const float* p = xyz + size_t(i) * stride_floats;
      if (!downsample_on) { insert(p); continue; }
      const float ds = ds_;
      int iv[3];
      float bmin[3], bmax[3], mid[3];
      for (int a = 0; a < 3; a++) {
        float f = std::floor(p[a] / ds);
        iv[a] = int(f);
        bmin[a] = f * ds;
        bmax[a] = bmin[a] + ds;
        mid[a] = float(bmin[a] + (bmax[a] - bmin[a]) / 2.0);
      }
      in_box.clear();
      box_query(iv, bmin, bmax, in_box);
      float min_dist = dist2(p, mid);
      int best = -1;  // -1: the new point itself
      for (int s : in_box) {
        float d = dist2(&xyz_[3 * size_t(s)], mid);
        if (d < min_dist) { min_dist = d; best = s; }
      }
      // same_point(PointToAdd[i], downsample_result): true when the result IS the new point, and also when
      // an existing point within 1e-6 of it won
      bool same = (best < 0) || same_point(p, &xyz_[3 * size_t(best)]);
      if (in_box.size() > 1 || same) {
        float keep[3];
        const float* src = best < 0 ? p : &xyz_[3 * size_t(best)];
        keep[0] = src[0]; keep[1] = src[1]; keep[2] = src[2];
        for (int s : in_box) erase(s);
        insert(keep);
        counter++;
      }
    }
    return counter;
  }

  // Copies the valid points as float4 (x, y, z, bit-cast slot id); returns the count.
  int export_float4(float* out4) const {
    int k = 0;
    for (size_t s = 0; s < state_.size(); s++)
      if (state_[s]) {
        out4[4 * size_t(k) + 0] = xyz_[3 * s];
        out4[4 * size_t(k) + 1] = xyz_[3 * s + 1];
        out4[4 * size_t(k) + 2] = xyz_[3 * s + 2];
        uint32_t id = uint32_t(s);
        std::memcpy(&out4[4 * size_t(k) + 3], &id, 4);
        k++;
      }
    return k;
  }

 private:
  static float dist2(const float* a, const float* b) {
    float d = (a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]);
    return d;
  }
  static bool same_point(const float* a, const float* b) {
    return std::fabs(a[0] - b[0]) < 1e-6 && std::fabs(a[1] - b[1]) < 1e-6 && std::fabs(a[2] - b[2]) < 1e-6;
  }
  static uint64_t pack(const int iv[3]) {
    const int64_t B = 1 << 20;
    return (uint64_t(iv[2] + B) << 42) | (uint64_t(iv[1] + B) << 21) | uint64_t(iv[0] + B);
  }
  static uint32_t hash(uint64_t k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return uint32_t(k);
  }
  bool in_axis_box(float v, int i) const {
    float lo = float(i) * ds_;
    float hi = lo + ds_;
    return lo <= v && hi > v;
  }
  // nominal voxel of a stored point, and whether it is "regular" (see header)
  bool classify(const float* p, int iv[3]) const {
    bool regular = true;
    for (int a = 0; a < 3; a++) {
      iv[a] = int(std::floor(p[a] / ds_));
      if (!in_axis_box(p[a], iv[a]) || in_axis_box(p[a], iv[a] - 1) || in_axis_box(p[a], iv[a] + 1)) regular = false;
    }
    return regular;
  }

  int* bucket_head(uint64_t key, bool create) {
    if ((n_buckets_ + 1) * 2 > int(table_.size())) grow();
    uint32_t mask = uint32_t(table_.size() - 1);
    uint32_t s = hash(key) & mask;
    while (true) {
      if (table_[s] == -1 && tkeys_[s] == 0) {  // never-used slot
        if (!create) return nullptr;
        tkeys_[s] = key + 1;  // keys are stored +1 so that 0 means "unused"
        n_buckets_++;
        return &table_[s];
      }
      if (tkeys_[s] == key + 1) return &table_[s];
      s = (s + 1) & mask;
    }
  }
  void grow() {
    std::vector<int> old_t;
    std::vector<uint64_t> old_k;
    old_t.swap(table_);
    old_k.swap(tkeys_);
    table_.assign(old_t.size() * 2, -1);
    tkeys_.assign(old_t.size() * 2, 0);
    n_buckets_ = 0;
    for (size_t i = 0; i < old_t.size(); i++)
      if (old_k[i] != 0 && old_t[i] != -1) *bucket_head(old_k[i] - 1, true) = old_t[i];
  }

  void insert(const float* p) {
    int s;
    if (!free_.empty()) { s = free_.back(); free_.pop_back(); }
    else {
      s = int(state_.size());
      xyz_.resize(xyz_.size() + 3);
      next_.push_back(-1);
      key_.push_back(0);
      state_.push_back(0);
    }
    xyz_[3 * size_t(s)] = p[0]; xyz_[3 * size_t(s) + 1] = p[1]; xyz_[3 * size_t(s) + 2] = p[2];
    int iv[3];
    if (classify(p, iv)) {
      uint64_t k = pack(iv);
      int* head = bucket_head(k, true);
      next_[s] = *head;
      *head = s;
      key_[s] = k;
      state_[s] = 1;
    } else {
      ambiguous_.push_back(s);
      state_[s] = 2;
    }
    n_alive_++;
    version_++;
  }
  void erase(int s) {
    if (state_[s] == 1) {
      int* head = bucket_head(key_[s], false);
      int* link = head;
      while (*link != s) link = &next_[*link];
      *link = next_[s];
    } else if (state_[s] == 2) {
      for (size_t i = 0; i < ambiguous_.size(); i++)
        if (ambiguous_[i] == s) { ambiguous_[i] = ambiguous_.back(); ambiguous_.pop_back(); break; }
    }
    state_[s] = 0;
    free_.push_back(s);
    n_alive_--;
    version_++;
  }
  void box_query(const int iv[3], const float bmin[3], const float bmax[3], std::vector<int>& out) {
    int* head = bucket_head(pack(iv), false);
    if (head)
      for (int s = *head; s != -1; s = next_[s]) out.push_back(s);  // regular points of this voxel: all inside
    for (int s : ambiguous_) {
      const float* q = &xyz_[3 * size_t(s)];
      if (bmin[0] <= q[0] && bmax[0] > q[0] && bmin[1] <= q[1] && bmax[1] > q[1] && bmin[2] <= q[2] && bmax[2] > q[2])
        out.push_back(s);
    }
  }

  float ds_ = 0.2f;  // KD_TREE default box_length (ikd_Tree.h:166); laserMapping sets filter_size_map (:924)
  std::vector<float> xyz_;
  std::vector<int> next_;
  std::vector<uint64_t> key_;
  std::vector<uint8_t> state_;  // 0 free, 1 bucketed, 2 ambiguous
  std::vector<int> free_;
  std::vector<int> ambiguous_;
  std::vector<int> table_;       // bucket heads
  std::vector<uint64_t> tkeys_;  // bucket keys + 1
  int n_buckets_ = 0;
  int n_alive_ = 0;
  uint64_t version_ = 0;
};

}  // namespace lii
