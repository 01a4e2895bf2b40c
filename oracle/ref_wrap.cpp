// ORACLE — TEST INFRASTRUCTURE ONLY.
// extern "C" wrapper around the UNMODIFIED reference k-d tree (compiled from
// /root/reference/include/ikd-Tree/ikd_Tree.cpp where it lies; never copied into this repo).
// Output: oracle/_ref/libref_ikdtree.so (git-ignored).  Used to pin the oracle's restated tree and,
// optionally, as the "reference" CPU baseline for the k-NN stage.
#include <unistd.h>

#include "ikd_Tree.h"

extern "C" {

void* ref_tree_create() { return new KD_TREE(); }  // ~100 MB (MANUAL_Q logger) — heap, never stack
void ref_tree_destroy(void* h) { delete static_cast<KD_TREE*>(h); }
void ref_tree_set_downsample(void* h, float box) { static_cast<KD_TREE*>(h)->set_downsample_param(box); }

static PointVector to_pv(const float* xyz, int n) {
  PointVector v(n);
  for (int i = 0; i < n; i++) {
    v[i].x = xyz[3 * i];
    v[i].y = xyz[3 * i + 1];
    v[i].z = xyz[3 * i + 2];
  }
  return v;
}

void ref_tree_build(void* h, const float* xyz, int n) { static_cast<KD_TREE*>(h)->Build(to_pv(xyz, n)); }

int ref_tree_add_points(void* h, const float* xyz, int n, int downsample_on) {
  PointVector v = to_pv(xyz, n);
  return static_cast<KD_TREE*>(h)->Add_Points(v, downsample_on != 0);
}
int ref_tree_delete_boxes(void* h, const float* boxes, int n) {  // n x 6: min xyz, max xyz
  std::vector<BoxPointType> v(n);
  for (int i = 0; i < n; i++)
    for (int a = 0; a < 3; a++) { v[i].vertex_min[a] = boxes[6 * i + a]; v[i].vertex_max[a] = boxes[6 * i + 3 + a]; }
  return static_cast<KD_TREE*>(h)->Delete_Point_Boxes(v);
}
int ref_tree_size(void* h) { return static_cast<KD_TREE*>(h)->size(); }
int ref_tree_validnum(void* h) { return static_cast<KD_TREE*>(h)->validnum(); }

int ref_tree_flatten(void* h, float* out_xyz, int cap, int settle_ms) {
  KD_TREE* t = static_cast<KD_TREE*>(h);
  if (settle_ms > 0) usleep(1000 * settle_ms);  // let the background rebuild thread finish
  PointVector v;
  t->flatten(t->Root_Node, v, NOT_RECORD);
  int n = int(v.size());
  for (int i = 0; i < n && i < cap; i++) {
    out_xyz[3 * i] = v[i].x;
    out_xyz[3 * i + 1] = v[i].y;
    out_xyz[3 * i + 2] = v[i].z;
  }
  return n;
}

void ref_tree_knn(void* h, const float* q, int nq, int k, double max_dist, float* out_pts, float* out_d2, int* out_n,
                  int threads) {
  KD_TREE* t = static_cast<KD_TREE*>(h);
#pragma omp parallel for num_threads(threads) schedule(static)
  for (int i = 0; i < nq; i++) {
    PointType p;
    p.x = q[3 * i];
    p.y = q[3 * i + 1];
    p.z = q[3 * i + 2];
    PointVector near;
    std::vector<float> d2;
    t->Nearest_Search(p, k, near, d2, max_dist);
    int f = int(near.size());
    out_n[i] = f;
    for (int j = 0; j < k; j++) {
      float* o = out_pts + 3 * (size_t(i) * k + j);
      if (j < f) {
        o[0] = near[j].x; o[1] = near[j].y; o[2] = near[j].z;
        out_d2[size_t(i) * k + j] = d2[j];
      } else {
        o[0] = o[1] = o[2] = 0.f;
        out_d2[size_t(i) * k + j] = INFINITY;
      }
    }
  }
}
}
