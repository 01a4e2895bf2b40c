// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.hpp header).
//
// Single-threaded, behaviour-compatible restatement of the reference's incremental k-d tree
// (include/ikd-Tree/ikd_Tree.{h,cpp}) for the operations the hot path uses:
//   Build ............... ikd_Tree.cpp:336-347, BuildTree :536-584 (longest-axis median split)
//   Nearest_Search ...... ikd_Tree.cpp:349-379, Search :825-968 (box-distance pruning, bounded
//                         max-heap, acceptance d2 <= max_dist, prune box_d2 > max_dist^2 — quirk A5)
//   Add_Points .......... ikd_Tree.cpp:381-456 (per-voxel "keep the point closest to the voxel
//                         centre" down-sampling), Add_by_point :775-823
//   Delete_by_range ..... ikd_Tree.cpp:608-670, Search_by_range :970-998
//   Update .............. ikd_Tree.cpp:1092-1227, Criterion_Check :1000-1017, Rebuild :586-606
//   calc_dist / calc_box_dist :1273-1289, MANUAL_HEAP :1296-1362, PointType_CMP ikd_Tree.h:50-61
// Deviations (results are unaffected — exact k-NN depends only on the valid point set):
//   * no background rebuild pthread / operation logger: every rebuild is synchronous;
//   * box deletions mark nodes eagerly instead of lazily pushing `tree_deleted` flags down.
// Pinned against the UNMODIFIED reference tree compiled out-of-tree (oracle/_ref, see Makefile)
// in tests/test_oracle_core.py::test_kdtree_against_reference_ikdtree (and, through the device map, in tests/test_gpu_map.py):
// identical neighbour sets and distances, identical tree contents after Add_Points / Delete_Point_Boxes streams.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

namespace orc {

struct KPoint {
  float x, y, z;
  int32_t id;  // caller tag (not used by the algorithm)
};

struct KdNode {
  KPoint p;
  int axis = 0;
  int tree_size = 1;
  int invalid_num = 0;
  bool point_deleted = false;
  bool tree_deleted = false;
  float rmin[3], rmax[3];
  KdNode* l = nullptr;
  KdNode* r = nullptr;
};

class KdTree {
 public:
  ~KdTree() { destroy(root_); }
  void set_downsample_param(float box) { downsample_size_ = box; }
  int size() const { return root_ ? root_->tree_size : 0; }
  int validnum() const { return root_ ? root_->tree_size - root_->invalid_num : 0; }
  bool empty() const { return root_ == nullptr; }

  void build(std::vector<KPoint> pts) {
    destroy(root_);
    root_ = nullptr;
    if (pts.empty()) return;
    root_ = build_rec(pts, 0, int(pts.size()) - 1);
  }

  // Nearest_Search: returns neighbours sorted ascending by squared distance.
  int nearest_search(const float q[3], int k, KPoint* out_pts, float* out_d2, double max_dist) const {
    Heap h(k);
    search(root_, k, q, h, max_dist);
    int found = h.n;
    // pop largest-first into the back of the output (ikd_Tree.cpp:372-377)
    for (int i = found - 1; i >= 0; i--) {
      out_pts[i] = h.e[0].p;
      out_d2[i] = h.e[0].d;
      h.pop();
    }
    return found;
  }

  // Add_Points(PointToAdd, downsample_on) — ikd_Tree.cpp:381-456.  Returns the reference's counter.
  int add_points(const std::vector<KPoint>& pts, bool downsample_on) {
    int counter = 0;
    for (const KPoint& pt : pts) {
      if (downsample_on) {
        const float ds = downsample_size_;
        float bmin[3], bmax[3];
        const float c[3] = {pt.x, pt.y, pt.z};
        KPoint mid = pt;
        float m[3];
        for (int a = 0; a < 3; a++) {
          bmin[a] = std::floor(c[a] / ds) * ds;
          bmax[a] = bmin[a] + ds;
          m[a] = bmin[a] + (bmax[a] - bmin[a]) / 2.0;
        }
        mid.x = m[0]; mid.y = m[1]; mid.z = m[2];
        std::vector<KPoint> in_box;
        search_by_range(root_, bmin, bmax, in_box);
        float min_dist = calc_dist(pt, mid);
        KPoint result = pt;
        for (const KPoint& s : in_box) {
          float d = calc_dist(s, mid);
          if (d < min_dist) { min_dist = d; result = s; }
        }
        if (in_box.size() > 1 || same_point(pt, result)) {
          if (!in_box.empty()) delete_by_range(&root_, bmin, bmax, true);
          add_by_point(&root_, result, true, root_ ? root_->axis : 0);
          counter++;
        }
      } else {
        add_by_point(&root_, pt, true, root_ ? root_->axis : 0);
      }
    }
    return counter;
  }

  void flatten(std::vector<KPoint>& out) const { flatten_rec(root_, out); }

 private:
  struct HeapE { KPoint p; float d; };
  // MANUAL_HEAP restated as a max-heap on (dist, then x) — ikd_Tree.h:50-61, ikd_Tree.cpp:1296-1362
  struct Heap {
    std::vector<HeapE> e;
    int n = 0;
    explicit Heap(int k) : e(size_t(2 * k)) {}
    static bool less(const HeapE& a, const HeapE& b) {
      if (std::fabs(a.d - b.d) < 1e-10) return a.p.x < b.p.x;
      return a.d < b.d;
    }
    void push(const HeapE& v) {
      if (n >= int(e.size())) return;
      int i = n++;
      e[i] = v;
      while (i > 0) {
        int a = (i - 1) / 2;
        if (less(e[a], v)) { e[i] = e[a]; i = a; } else break;
      }
      e[i] = v;
    }
    void pop() {
      if (n == 0) return;
      e[0] = e[n - 1];
      n--;
      int i = 0, l = 1;
      HeapE tmp = e[0];
      while (l < n) {
        if (l + 1 < n && less(e[l], e[l + 1])) l++;
        if (less(tmp, e[l])) { e[i] = e[l]; i = l; l = 2 * i + 1; } else break;
      }
      e[i] = tmp;
    }
  };

  static float calc_dist(const KPoint& a, const KPoint& b) {
    float d = (a.x - b.x) * (a.x - b.x) + (a.y - b.y) * (a.y - b.y) + (a.z - b.z) * (a.z - b.z);
    return d;
  }
  static float calc_dist_q(const float q[3], const KPoint& b) {
    float d = (q[0] - b.x) * (q[0] - b.x) + (q[1] - b.y) * (q[1] - b.y) + (q[2] - b.z) * (q[2] - b.z);
    return d;
  }
  static float box_dist(const KdNode* n, const float q[3]) {
    if (n == nullptr) return INFINITY;
    float md = 0.0f;
    for (int a = 0; a < 3; a++) {
      if (q[a] < n->rmin[a]) md += (q[a] - n->rmin[a]) * (q[a] - n->rmin[a]);
      if (q[a] > n->rmax[a]) md += (q[a] - n->rmax[a]) * (q[a] - n->rmax[a]);
    }
    return md;
  }
  static bool same_point(const KPoint& a, const KPoint& b) {
    return std::fabs(a.x - b.x) < 1e-6 && std::fabs(a.y - b.y) < 1e-6 && std::fabs(a.z - b.z) < 1e-6;
  }
  static float coord(const KPoint& p, int a) { return a == 0 ? p.x : (a == 1 ? p.y : p.z); }

  void destroy(KdNode* n) {
    if (!n) return;
    destroy(n->l);
    destroy(n->r);
    delete n;
  }

  // Update — ikd_Tree.cpp:1092-1227 (sizes, invalid counts, tree_deleted, tight ranges)
  static void update(KdNode* n) {
    KdNode* ch[2] = {n->l, n->r};
    n->tree_size = 1;
    n->invalid_num = n->point_deleted ? 1 : 0;
    bool all_del = n->point_deleted;
    bool none_del = !n->point_deleted;
    for (KdNode* c : ch)
      if (c) {
        n->tree_size += c->tree_size;
        n->invalid_num += c->invalid_num;
        all_del = all_del && c->tree_deleted;
        none_del = none_del && !c->tree_deleted;
      }
    n->tree_deleted = all_del;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    const bool take_all = n->tree_deleted || none_del;
    for (KdNode* c : ch)
      if (c && (take_all || !c->tree_deleted))
        for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], c->rmin[a]); hi[a] = std::max(hi[a], c->rmax[a]); }
    if (take_all || !n->point_deleted)
      for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], coord(n->p, a)); hi[a] = std::max(hi[a], coord(n->p, a)); }
    for (int a = 0; a < 3; a++) { n->rmin[a] = lo[a]; n->rmax[a] = hi[a]; }
  }

  // BuildTree — ikd_Tree.cpp:536-584
  KdNode* build_rec(std::vector<KPoint>& s, int l, int r) {
    if (l > r) return nullptr;
    KdNode* n = new KdNode;
    int mid = (l + r) >> 1;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = l; i <= r; i++)
      for (int a = 0; a < 3; a++) { mn[a] = std::min(mn[a], coord(s[i], a)); mx[a] = std::max(mx[a], coord(s[i], a)); }
    int axis = 0;
    float range[3];
    for (int a = 0; a < 3; a++) range[a] = mx[a] - mn[a];
    for (int a = 1; a < 3; a++) if (range[a] > range[axis]) axis = a;
    n->axis = axis;
    std::nth_element(s.begin() + l, s.begin() + mid, s.begin() + r + 1,
                     [axis](const KPoint& a, const KPoint& b) { return coord(a, axis) < coord(b, axis); });
    n->p = s[mid];
    n->l = build_rec(s, l, mid - 1);
    n->r = build_rec(s, mid + 1, r);
    update(n);
    return n;
  }

  // Criterion_Check — ikd_Tree.cpp:1000-1017 (delete 0.5, balance 0.6 from the KD_TREE ctor defaults,
  // ikd_Tree.h:166; laserMapping's global `KD_TREE ikdtree;` uses them)
  static bool criterion(const KdNode* n) {
    if (n->tree_size <= 10) return false;
    const KdNode* son = n->l ? n->l : n->r;
    float del_eval = float(n->invalid_num) / n->tree_size;
    float bal_eval = float(son->tree_size) / (n->tree_size - 1);
    if (del_eval > 0.5f) return true;
    if (bal_eval > 0.6f || bal_eval < 1 - 0.6f) return true;
    return false;
  }
  // Rebuild — ikd_Tree.cpp:586-606 (always synchronous here)
  void rebuild(KdNode** n) {
    std::vector<KPoint> pts;
    pts.reserve((*n)->tree_size);
    flatten_rec(*n, pts);
    destroy(*n);
    *n = pts.empty() ? nullptr : build_rec(pts, 0, int(pts.size()) - 1);
  }

  // Add_by_point — ikd_Tree.cpp:775-823
  void add_by_point(KdNode** n, const KPoint& pt, bool allow_rebuild, int father_axis) {
    if (*n == nullptr) {
      *n = new KdNode;
      (*n)->p = pt;
      (*n)->axis = (father_axis + 1) % 3;
      update(*n);
      return;
    }
    KdNode* cur = *n;
    if (coord(pt, cur->axis) < coord(cur->p, cur->axis))
      add_by_point(&cur->l, pt, allow_rebuild, cur->axis);
    else
      add_by_point(&cur->r, pt, allow_rebuild, cur->axis);
    update(cur);
    if (allow_rebuild && criterion(cur)) rebuild(n);
  }

  // Delete_by_range(is_downsample) — ikd_Tree.cpp:608-670 (eager marking, see header)
  int delete_by_range(KdNode** n, const float bmin[3], const float bmax[3], bool allow_rebuild) {
    KdNode* cur = *n;
    if (cur == nullptr || cur->tree_deleted) return 0;
    for (int a = 0; a < 3; a++)
      if (bmax[a] <= cur->rmin[a] || bmin[a] > cur->rmax[a]) return 0;
    int cnt = 0;
    if (!cur->point_deleted && in_box(cur->p, bmin, bmax)) {
      cur->point_deleted = true;
      cnt++;
    }
    cnt += delete_by_range(&cur->l, bmin, bmax, allow_rebuild);
    cnt += delete_by_range(&cur->r, bmin, bmax, allow_rebuild);
    update(cur);
    if (allow_rebuild && criterion(cur)) rebuild(n);
    return cnt;
  }
  static bool in_box(const KPoint& p, const float bmin[3], const float bmax[3]) {
    return bmin[0] <= p.x && bmax[0] > p.x && bmin[1] <= p.y && bmax[1] > p.y && bmin[2] <= p.z && bmax[2] > p.z;
  }
  // Search_by_range — ikd_Tree.cpp:970-998
  void search_by_range(const KdNode* n, const float bmin[3], const float bmax[3], std::vector<KPoint>& out) const {
    if (n == nullptr) return;
    for (int a = 0; a < 3; a++)
      if (bmax[a] <= n->rmin[a] || bmin[a] > n->rmax[a]) return;
    if (!n->point_deleted && in_box(n->p, bmin, bmax)) out.push_back(n->p);
    search_by_range(n->l, bmin, bmax, out);
    search_by_range(n->r, bmin, bmax, out);
  }
  void flatten_rec(const KdNode* n, std::vector<KPoint>& out) const {
    if (!n) return;
    if (!n->point_deleted) out.push_back(n->p);
    flatten_rec(n->l, out);
    flatten_rec(n->r, out);
  }

  // Search — ikd_Tree.cpp:825-968
  void search(const KdNode* n, int k, const float q[3], Heap& h, double max_dist) const {
    if (n == nullptr || n->tree_deleted) return;
    double cur = box_dist(n, q);
    if (cur > max_dist * max_dist) return;
    if (!n->point_deleted) {
      float d = calc_dist_q(q, n->p);
      if (d <= max_dist && (h.n < k || d < h.e[0].d)) {
        if (h.n >= k) h.pop();
        h.push(HeapE{n->p, d});
      }
    }
    float dl = box_dist(n->l, q);
    float dr = box_dist(n->r, q);
    if (h.n < k || (dl < h.e[0].d && dr < h.e[0].d)) {
      if (dl <= dr) {
        search(n->l, k, q, h, max_dist);
        if (h.n < k || dr < h.e[0].d) search(n->r, k, q, h, max_dist);
      } else {
        search(n->r, k, q, h, max_dist);
        if (h.n < k || dl < h.e[0].d) search(n->l, k, q, h, max_dist);
      }
    } else {
      if (dl < h.e[0].d) search(n->l, k, q, h, max_dist);
      if (dr < h.e[0].d) search(n->r, k, q, h, max_dist);
    }
  }

  KdNode* root_ = nullptr;
  float downsample_size_ = 0.2f;
};

}  // namespace orc
