// ORACLE - TEST INFRASTRUCTURE ONLY.
// C entry points around the reference's OWN TEXT of the per-scan update: the pieces oracle/ref_slice_iekf.py cuts out of
// /root/reference/src/laserMapping.cpp at build time (oracle/_ref/gen/*.inc, git-ignored; nothing is copied into the repository) -
// the file-scope variables the update uses, calc_dist, calcBodyVar, pointBodyToWorld (:153-220), map_incremental (:516-559), the
// declarations and initialisation of main()'s solver matrices (:808-816, :841-843) and the iterated update itself (:936-1134) -
// compiled together with the UNMODIFIED include/common_lib.h (esti_plane<double>, StatesGroup), include/so3_math.h and
// include/ikd-Tree/ikd_Tree.cpp into oracle/_ref/libref_iekf.so, against oracle/ref_shim_iekf + oracle/ref_shim_math (see the header
// of ref_shim_iekf/Eigen/Core for what the shim supplies and what it therefore does NOT pin: the inside of Eigen's inverse(),
// ColPivHouseholderQR and matrix products).  tests/test_oracle_iekf_pinned.py holds oracle/orc_iekf.hpp to this library.
// The code below only moves data between flat arrays and the reference's variables and calls the slices.
#include <omp.h>
#include <unistd.h>

#include <cmath>
#include <condition_variable>
#include <deque>
#include <fstream>
#include <iostream>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include <common_lib.h>
#include <geometry_msgs/Quaternion.h>
#include <tf/transform_datatypes.h>
#include <ikd-Tree/ikd_Tree.h>

// ---- the reference's text ------------------------------------------------------------------------------------------------------
#include "gen/globals.inc"
#include "gen/calc_dist.inc"
#include "gen/calc_body_var.inc"
#include "gen/point_body_to_world.inc"
#include "gen/map_incremental.inc"
// main()'s locals in front of its loop: declared once, like there
#include "gen/main_decls.inc"
static void ref_main_init() {
#include "gen/main_init.inc"
}
static int ref_main_update() {  // one pass of main()'s loop body, from "ICP and iterated Kalman filter update" to the end of the iteration loop
#include "gen/main_update.inc"
  return rematch_num;
}
// ---------------------------------------------------------------------------------------------------------------------------------

namespace {
bool g_init = false;
void from_pod(StatesGroup& s, const double* p, bool with_cov) {
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { s.rot_end(i, j) = p[3 * i + j]; s.offset_R_L_I(i, j) = p[12 + 3 * i + j]; }
  for (int i = 0; i < 3; i++) { s.pos_end(i) = p[9 + i]; s.offset_T_L_I(i) = p[21 + i]; s.vel_end(i) = p[24 + i]; s.bias_g(i) = p[27 + i]; s.bias_a(i) = p[30 + i]; s.gravity(i) = p[33 + i]; }
  if (with_cov) for (int i = 0; i < DIM_STATE; i++) for (int j = 0; j < DIM_STATE; j++) s.cov(i, j) = p[36 + DIM_STATE * i + j];
}
void to_pod(const StatesGroup& s, double* p) {
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { p[3 * i + j] = s.rot_end(i, j); p[12 + 3 * i + j] = s.offset_R_L_I(i, j); }
  for (int i = 0; i < 3; i++) { p[9 + i] = s.pos_end(i); p[21 + i] = s.offset_T_L_I(i); p[24 + i] = s.vel_end(i); p[27 + i] = s.bias_g(i); p[30 + i] = s.bias_a(i); p[33 + i] = s.gravity(i); }
  for (int i = 0; i < DIM_STATE; i++) for (int j = 0; j < DIM_STATE; j++) p[36 + DIM_STATE * i + j] = s.cov(i, j);
}
}  // namespace

extern "C" {
// the local map: ikdtree.set_downsample_param + Build, as main() does with the first scan (src/laserMapping.cpp:923-929)
int ref_iekf_map_build(const float* xyz, int n, double filter_size_map) {
  filter_size_map_min = filter_size_map;
  ikdtree.set_downsample_param(filter_size_map_min);
  PointVector v(n);
  for (int i = 0; i < n; i++) { v[i].x = xyz[3 * i]; v[i].y = xyz[3 * i + 1]; v[i].z = xyz[3 * i + 2]; }
  ikdtree.Build(v);
  return ikdtree.size();
}
// One scan through the reference's update.  body: n x 3 (the down-sampled scan, LiDAR frame); state: lii_state POD (612 doubles), in:
// the propagated state, out: the updated state; the propagated state of the update is the input state (main(): state_propagat = state).
// selected / normvec (n x 4: normal, pd2) / near (n x 5 x 3) / near_n: the reference's arrays after the last iteration.
int ref_iekf_update(const float* body, int n, double* state_pod, int max_iterations, int imu_enabled, int* iterations, int* rematch,
                    int* effect_num, unsigned char* selected, float* normvec_out, float* near, int* near_n) {
  if (n > 100000) return -1;  // quirk A1: the reference's arrays end there
  if (!g_init) { ref_main_init(); memset(point_selected_surf, true, sizeof(point_selected_surf)); g_init = true; }
  from_pod(state, state_pod, true);
  state_propagat = state;
  NUM_MAX_ITERATIONS = max_iterations;
  imu_en = imu_enabled != 0;
  feats_down_body->points.resize(n);
  for (int i = 0; i < n; i++) {
    PointType p;
    p.x = body[3 * i]; p.y = body[3 * i + 1]; p.z = body[3 * i + 2];
    feats_down_body->points[i] = p;
  }
  feats_down_size = n;
  const int rm = ref_main_update();
  to_pod(state, state_pod);
  if (iterations) *iterations = iterCount + 1;  // (the loop always leaves through its `break`)
  if (rematch) *rematch = rm;
  if (effect_num) *effect_num = effect_feat_num;
  for (int i = 0; i < n; i++) {
    if (selected) selected[i] = point_selected_surf[i] ? 1 : 0;
    if (normvec_out) { normvec_out[4 * i] = normvec->points[i].x; normvec_out[4 * i + 1] = normvec->points[i].y; normvec_out[4 * i + 2] = normvec->points[i].z; normvec_out[4 * i + 3] = normvec->points[i].intensity; }
    const PointVector& pn = Nearest_Points[i];
    if (near_n) near_n[i] = int(pn.size());
    if (near) for (int k = 0; k < 5; k++) for (int a = 0; a < 3; a++) near[15 * i + 3 * k + a] = k < int(pn.size()) ? (a == 0 ? pn[k].x : (a == 1 ? pn[k].y : pn[k].z)) : 0.f;
  }
  return 0;
}
// map_incremental() on the scan the last ref_iekf_update registered; returns add_point_size (:558), *tree_size = ikdtree.size()
int ref_iekf_map_incremental(int* tree_size) {
  map_incremental();
  if (tree_size) *tree_size = ikdtree.size();
  return add_point_size;
}
// esti_plane<double> (include/common_lib.h:236-269) on five neighbours (x, y, z as float): the instantiation main() uses (:997)
int ref_esti_plane(const float* pts15, double threshold, double* pabcd) {
  PointVector pv(5);
  for (int j = 0; j < 5; j++) { pv[j].x = pts15[3 * j]; pv[j].y = pts15[3 * j + 1]; pv[j].z = pts15[3 * j + 2]; }
  VD(4) r;
  r.setZero();
  const bool ok = esti_plane(r, pv, threshold);
  for (int k = 0; k < 4; k++) pabcd[k] = r(k);
  return ok ? 1 : 0;
}
int ref_iekf_tree_flatten(float* out_xyz, int cap, int settle_ms) {
  if (settle_ms > 0) usleep(1000 * settle_ms);
  PointVector v;
  ikdtree.flatten(ikdtree.Root_Node, v, NOT_RECORD);
  const int n = int(v.size());
  for (int i = 0; i < n && i < cap; i++) { out_xyz[3 * i] = v[i].x; out_xyz[3 * i + 1] = v[i].y; out_xyz[3 * i + 2] = v[i].z; }
  return n;
}
}
