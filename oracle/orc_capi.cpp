// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.hpp header).
// extern "C" surface of the CPU restatement, loaded by oracle/oracle.py through ctypes.
// Every entry point forwards to a restated reference function; the citation lives with the callee.
#include <chrono>
#include <cstring>

#include "orc_iekf.hpp"
#include "orc_kdtree.hpp"
#include "orc_math.hpp"
#include "orc_plane.hpp"
#include "orc_scan.hpp"
#include "orc_ingest.hpp"

#ifdef _OPENMP
#include <omp.h>
#endif

using namespace orc;

namespace {
struct TreeBox {
  KdTree tree;
  IekfScratch scratch;  // holds Nearest_Points etc. of the last update (for map_incremental)
};
std::vector<KPoint> to_kpoints(const float* xyz, int n, int id0 = 0) {
  std::vector<KPoint> v(n);
  for (int i = 0; i < n; i++) v[i] = KPoint{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], id0 + i};
  return v;
}
std::vector<P4> to_p4(const float* p, int n) {
  std::vector<P4> v(n);
  std::memcpy(v.data(), p, sizeof(P4) * size_t(n));
  return v;
}
}  // namespace

extern "C" {

int orc_num_procs() {
#ifdef _OPENMP
  return omp_get_num_procs();
#else
  return 1;
#endif
}

// ---- SO(3) / state ------------------------------------------------------------------------
void orc_exp(const double* w, double dt, double* R9) {
  M3 R = Exp(V3(w[0], w[1], w[2]), dt);
  std::memcpy(R9, R.m, 72);
}
void orc_exp1(const double* w, double* R9) {
  M3 R = Exp(V3(w[0], w[1], w[2]));
  std::memcpy(R9, R.m, 72);
}
void orc_exp3(double a, double b, double c, double* R9) {
  M3 R = Exp3(a, b, c);
  std::memcpy(R9, R.m, 72);
}
void orc_log(const double* R9, double* out3) {
  V3 v = Log(M3::from(R9));
  out3[0] = v.x; out3[1] = v.y; out3[2] = v.z;
}
void orc_rot_to_euler(const double* R9, double* out3) {
  V3 v = RotMtoEuler(M3::from(R9));
  out3[0] = v.x; out3[1] = v.y; out3[2] = v.z;
}
int orc_state_doubles() { return STATE_DOUBLES; }
void orc_state_init(double* pod) {
  State s;
  state_to_pod(s, pod);
}
void orc_state_boxplus(double* pod, const double* d24) {
  State s = state_from_pod(pod);
  boxplus(s, d24);
  state_to_pod(s, pod);
}
void orc_state_boxminus(const double* a, const double* b, double* out24) {
  boxminus(state_from_pod(a), state_from_pod(b), out24);
}
void orc_inverse(const double* A, int n, double* out) {
  Mat m(n, n);
  std::memcpy(m.a.data(), A, sizeof(double) * size_t(n) * n);
  Mat r = inverse(m);
  std::memcpy(out, r.a.data(), sizeof(double) * size_t(n) * n);
}

// ---- plane fit -----------------------------------------------------------------------------
int orc_esti_plane(const float* pts15, double threshold, double* pabcd) {
  return esti_plane(pabcd, pts15, threshold) ? 1 : 0;
}
void orc_esti_plane_batch(const float* pts15, int n, double threshold, double* pabcd, unsigned char* valid) {
  for (int i = 0; i < n; i++) valid[i] = esti_plane(pabcd + 4 * size_t(i), pts15 + 15 * size_t(i), threshold) ? 1 : 0;
}

// ---- k-d tree ------------------------------------------------------------------------------
void* orc_tree_create() { return new TreeBox; }
void orc_tree_destroy(void* h) { delete static_cast<TreeBox*>(h); }
void orc_tree_set_downsample(void* h, float box) { static_cast<TreeBox*>(h)->tree.set_downsample_param(box); }
void orc_tree_build(void* h, const float* xyz, int n) { static_cast<TreeBox*>(h)->tree.build(to_kpoints(xyz, n)); }
int orc_tree_add_points(void* h, const float* xyz, int n, int downsample_on) {
  return static_cast<TreeBox*>(h)->tree.add_points(to_kpoints(xyz, n), downsample_on != 0);
}
int orc_tree_size(void* h) { return static_cast<TreeBox*>(h)->tree.size(); }
int orc_tree_validnum(void* h) { return static_cast<TreeBox*>(h)->tree.validnum(); }
int orc_tree_flatten(void* h, float* out_xyz, int cap) {
  std::vector<KPoint> v;
  static_cast<TreeBox*>(h)->tree.flatten(v);
  int n = int(v.size());
  for (int i = 0; i < n && i < cap; i++) {
    out_xyz[3 * i] = v[i].x; out_xyz[3 * i + 1] = v[i].y; out_xyz[3 * i + 2] = v[i].z;
  }
  return n;
}
void orc_tree_knn(void* h, const float* q, int nq, int k, double max_dist, float* out_pts, float* out_d2, int* out_n,
                  int threads) {
  const KdTree& t = static_cast<TreeBox*>(h)->tree;
#ifdef _OPENMP
#pragma omp parallel for num_threads(threads) schedule(static)
#endif
  for (int i = 0; i < nq; i++) {
    std::vector<KPoint> np(k);
    std::vector<float> nd(k);
    int f = t.nearest_search(q + 3 * size_t(i), k, np.data(), nd.data(), max_dist);
    out_n[i] = f;
    for (int j = 0; j < k; j++) {
      float* o = out_pts + 3 * (size_t(i) * k + j);
      if (j < f) { o[0] = np[j].x; o[1] = np[j].y; o[2] = np[j].z; out_d2[size_t(i) * k + j] = nd[j]; }
      else { o[0] = o[1] = o[2] = 0.f; out_d2[size_t(i) * k + j] = INFINITY; }
    }
  }
}

// ---- undistortion / voxel grid ------------------------------------------------------------------
void orc_sort_by_time(float* pts4, int n) {
  std::vector<P4> v = to_p4(pts4, n);
  sort_by_time(v);
  std::memcpy(pts4, v.data(), sizeof(P4) * size_t(n));
}
void orc_undistort_imu(float* pts4, int n, const double* poses22, int K, const double* endR, const double* endp,
                       const double* RLI, const double* TLI) {
  std::vector<P4> v = to_p4(pts4, n);
  sort_by_time(v);
  std::vector<Pose6D> ps(K);
  std::memcpy(ps.data(), poses22, sizeof(Pose6D) * size_t(K));
  undistort_imu(v, ps, M3::from(endR), V3(endp[0], endp[1], endp[2]), M3::from(RLI), V3(TLI[0], TLI[1], TLI[2]));
  std::memcpy(pts4, v.data(), sizeof(P4) * size_t(n));
}
void orc_undistort_cv(float* pts4, int n, const double* omega, const double* vel, const double* endR) {
  std::vector<P4> v = to_p4(pts4, n);
  sort_by_time(v);
  undistort_cv(v, V3(omega[0], omega[1], omega[2]), V3(vel[0], vel[1], vel[2]), M3::from(endR));
  std::memcpy(pts4, v.data(), sizeof(P4) * size_t(n));
}
int orc_voxel_grid(const float* pts4, int n, float leaf, float* out4, int* out_n) {
  std::vector<P4> out;
  bool filtered = voxel_grid(to_p4(pts4, n), leaf, out);
  *out_n = int(out.size());
  std::memcpy(out4, out.data(), sizeof(P4) * out.size());
  return filtered ? 1 : 0;
}

// ---- IEKF ------------------------------------------------------------------------------------
// logs: per iteration 116 doubles = [searched, effect_num, HTH[78], HTz[12], solution[24]]
int orc_iekf_update(void* h, const float* body4, int n, double* state_pod, const double* state_prop_pod,
                    int max_iterations, int imu_en, int num_threads, int literal_gain, double* logs, int logs_cap,
                    float* out_nearest /*n*15*/, int* out_nearest_n, unsigned char* out_selected,
                    float* out_normvec /*n*4*/, float* out_world /*n*3*/, double* seconds) {
  TreeBox* tb = static_cast<TreeBox*>(h);
  std::vector<P4> body = to_p4(body4, n);
  State st = state_from_pod(state_pod), sp = state_from_pod(state_prop_pod);
  IekfParams prm;
  prm.max_iterations = max_iterations;
  prm.imu_en = imu_en;
  prm.num_threads = num_threads;
  prm.literal_gain = literal_gain;
  std::vector<IekfIterLog> lg;
  auto t0 = std::chrono::steady_clock::now();
  int iters = iekf_update(tb->tree, body, st, sp, prm, tb->scratch, lg);
  auto t1 = std::chrono::steady_clock::now();
  if (seconds) *seconds = std::chrono::duration<double>(t1 - t0).count();
  state_to_pod(st, state_pod);
  for (int i = 0; i < iters && i < logs_cap; i++) {
    double* o = logs + 116 * size_t(i);
    o[0] = lg[i].searched;
    o[1] = lg[i].effect_num;
    std::memcpy(o + 2, lg[i].HTH, 78 * 8);
    std::memcpy(o + 80, lg[i].HTz, 12 * 8);
    std::memcpy(o + 92, lg[i].solution, 24 * 8);
  }
  const IekfScratch& sc = tb->scratch;
  if (out_nearest) std::memcpy(out_nearest, sc.nearest.data(), sizeof(float) * sc.nearest.size());
  if (out_nearest_n) std::memcpy(out_nearest_n, sc.nearest_n.data(), sizeof(int32_t) * sc.nearest_n.size());
  if (out_selected) std::memcpy(out_selected, sc.selected.data(), sc.selected.size());
  if (out_normvec) std::memcpy(out_normvec, sc.normvec.data(), sizeof(float) * sc.normvec.size());
  if (out_world) std::memcpy(out_world, sc.world.data(), sizeof(float) * sc.world.size());
  return iters;
}

// One residual pass + normal equations at a FIXED state (no state update) — for per-kernel parity.
// out91 = [HTH upper 78, HTz 12, effect_num]
void orc_iekf_iterate_once(void* h, const float* body4, int n, const double* state_pod, int search, int imu_en,
                           int num_threads, double* out91, float* out_nearest, int* out_nearest_n,
                           unsigned char* inout_selected, float* out_normvec, double* out_pabcd) {
  TreeBox* tb = static_cast<TreeBox*>(h);
  std::vector<P4> body = to_p4(body4, n);
  State st = state_from_pod(state_pod);
  IekfParams prm;
  prm.imu_en = imu_en;
  prm.num_threads = num_threads;
  IekfScratch& sc = tb->scratch;
  if (search || int(sc.nearest_n.size()) != n) {
    sc.world.assign(3 * size_t(n), 0.f);
    sc.nearest.resize(15 * size_t(n));
    sc.nearest_d2.resize(5 * size_t(n));
    sc.nearest_n.assign(n, 0);
    sc.selected.assign(n, 1);
    sc.normvec.assign(4 * size_t(n), 0.f);
    sc.pabcd.assign(4 * size_t(n), 0.0);
    sc.res_last.assign(n, -1000.f);
  }
  if (!search && inout_selected) std::memcpy(sc.selected.data(), inout_selected, n);
  residual_pass(tb->tree, body, st, search != 0, prm, sc);
  double G[12][12] = {{0}}, g[12] = {0};
  int m = 0;
  for (int i = 0; i < n; i++)
    if (sc.selected[i]) {
      double hrow[12], z;
      const float pb[3] = {body[i].x, body[i].y, body[i].z};
      jacobian_row(st, pb, &sc.normvec[4 * size_t(i)], imu_en, hrow, z);
      for (int a = 0; a < 12; a++) {
        double ha = hrow[a] * prm.laser_point_cov_inv;
        g[a] += ha * z;
        for (int b = 0; b < 12; b++) G[a][b] += ha * hrow[b];
      }
      m++;
    }
  int t = 0;
  for (int a = 0; a < 12; a++)
    for (int b = a; b < 12; b++) out91[t++] = G[a][b];
  for (int a = 0; a < 12; a++) out91[78 + a] = g[a];
  out91[90] = m;
  if (out_nearest) std::memcpy(out_nearest, sc.nearest.data(), sizeof(float) * sc.nearest.size());
  if (out_nearest_n) std::memcpy(out_nearest_n, sc.nearest_n.data(), sizeof(int32_t) * sc.nearest_n.size());
  if (inout_selected) std::memcpy(inout_selected, sc.selected.data(), n);
  if (out_normvec) std::memcpy(out_normvec, sc.normvec.data(), sizeof(float) * sc.normvec.size());
  if (out_pabcd) std::memcpy(out_pabcd, sc.pabcd.data(), sizeof(double) * sc.pabcd.size());
}

// map_incremental using the Nearest_Points kept from the last orc_iekf_update on this tree.
// Returns counts through n_add / n_nodown; the selected world points are written to out_add / out_nodown (xyz).
void orc_map_incremental(void* h, const float* body4, int n, const double* state_pod, float filter_size_map,
                         int apply, float* out_add, int* n_add, float* out_nodown, int* n_nodown) {
  TreeBox* tb = static_cast<TreeBox*>(h);
  std::vector<P4> body = to_p4(body4, n);
  State st = state_from_pod(state_pod);
  std::vector<KPoint> a, b;
  map_incremental(tb->tree, body, st, tb->scratch, filter_size_map, a, b, apply != 0);
  *n_add = int(a.size());
  *n_nodown = int(b.size());
  for (size_t i = 0; i < a.size(); i++) { out_add[3 * i] = a[i].x; out_add[3 * i + 1] = a[i].y; out_add[3 * i + 2] = a[i].z; }
  for (size_t i = 0; i < b.size(); i++) { out_nodown[3 * i] = b[i].x; out_nodown[3 * i + 1] = b[i].y; out_nodown[3 * i + 2] = b[i].z; }
}

// ---- ingest (orc_ingest.hpp): frames are returned flattened; returns the number of frames or -1
static int flatten_frames(const std::vector<orc::Frame>& fr, float* out4, int cap_pts, double* begin_ms, int* offsets, int* counts,
                          int cap_frames) {
  int off = 0;
  if ((int)fr.size() > cap_frames) return -2;
  for (size_t k = 0; k < fr.size(); k++) {
    if (off + (int)fr[k].pts.size() > cap_pts) return -2;
    begin_ms[k] = fr[k].begin_time_ms;
    offsets[k] = off;
    counts[k] = (int)fr[k].pts.size();
    if (!fr[k].pts.empty()) std::memcpy(out4 + 4 * (size_t)off, fr[k].pts.data(), sizeof(orc::P4) * fr[k].pts.size());
    off += (int)fr[k].pts.size();
  }
  return (int)fr.size();
}
int orc_ingest_pcl2(const unsigned char* data, int n, const int* fields7, int lidar_type, int n_scans, int point_filter_num,
                    double blind, double stamp_s, int cut_frame_num, int scan_count, float* out4, int cap_pts,
                    double* begin_ms, int* offsets, int* counts, int cap_frames) {
  orc::Pc2Fields f{fields7[0], fields7[1], fields7[2], fields7[3], fields7[4], fields7[5], fields7[6]};
  orc::IngestOpts o{lidar_type, n_scans, point_filter_num, blind, stamp_s, cut_frame_num, scan_count};
  std::vector<orc::Frame> fr;
  if (orc::ingest_pcl2(data, n, f, o, fr) != 0) return -1;
  return flatten_frames(fr, out4, cap_pts, begin_ms, offsets, counts, cap_frames);
}
int orc_ingest_livox(const unsigned char* data, int n, const int* fields8, int n_scans, int point_filter_num, double blind,
                     double stamp_s, int cut_frame_num, int scan_count, float* out4, int cap_pts, double* begin_ms,
                     int* offsets, int* counts, int cap_frames) {
  orc::LivoxFields f{fields8[0], fields8[1], fields8[2], fields8[3], fields8[4], fields8[5], fields8[6], fields8[7]};
  orc::IngestOpts o{orc::AVIA, n_scans, point_filter_num, blind, stamp_s, cut_frame_num, scan_count};
  std::vector<orc::Frame> fr;
  if (orc::ingest_livox(data, n, f, o, fr) != 0) return -1;
  return flatten_frames(fr, out4, cap_pts, begin_ms, offsets, counts, cap_frames);
}

}  // extern "C"
