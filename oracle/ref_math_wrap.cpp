// C entry points around the UNMODIFIED reference headers include/so3_math.h and include/common_lib.h (StatesGroup), compiled
// where they lie under /root/reference against oracle/ref_shim_math (a minimal fixed-size matrix standing in for Eigen; ROS /
// PCL stand-ins) into oracle/_ref/libref_math.so.  Test infrastructure: tests/test_oracle_core.py holds oracle/orc_math.hpp to
// these functions bit for bit.  Nothing of the reference is copied here - the code below only converts between the flat
// 612-double state POD of the C-ABI and the reference's StatesGroup.
#include <common_lib.h>

namespace {
M3D m3(const double* r) { M3D m; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) m(i, j) = r[3 * i + j]; return m; }
void out3(const M3D& m, double* r) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r[3 * i + j] = m(i, j); }
V3D v3(const double* v) { return V3D(v[0], v[1], v[2]); }
void outv(const V3D& v, double* o) { for (int i = 0; i < 3; i++) o[i] = v(i); }
StatesGroup from_pod(const double* p) {
  StatesGroup s;
  s.rot_end = m3(p); s.pos_end = v3(p + 9); s.offset_R_L_I = m3(p + 12); s.offset_T_L_I = v3(p + 21);
  s.vel_end = v3(p + 24); s.bias_g = v3(p + 27); s.bias_a = v3(p + 30); s.gravity = v3(p + 33);
  for (int i = 0; i < DIM_STATE; i++) for (int j = 0; j < DIM_STATE; j++) s.cov(i, j) = p[36 + DIM_STATE * i + j];
  return s;
}
void to_pod(const StatesGroup& s, double* p) {
  out3(s.rot_end, p); outv(s.pos_end, p + 9); out3(s.offset_R_L_I, p + 12); outv(s.offset_T_L_I, p + 21);
  outv(s.vel_end, p + 24); outv(s.bias_g, p + 27); outv(s.bias_a, p + 30); outv(s.gravity, p + 33);
  for (int i = 0; i < DIM_STATE; i++) for (int j = 0; j < DIM_STATE; j++) p[36 + DIM_STATE * i + j] = s.cov(i, j);
}
}  // namespace

extern "C" {
void ref_exp1(const double* w, double* R) { out3(Exp(v3(w)), R); }
void ref_exp_dt(const double* w, double dt, double* R) { out3(Exp(v3(w), dt), R); }
void ref_exp3(double a, double b, double c, double* R) { out3(Exp(a, b, c), R); }
void ref_log(const double* R, double* o) { outv(Log(m3(R)), o); }
void ref_rot_to_euler(const double* R, double* o) { outv(RotMtoEuler(m3(R)), o); }
void ref_skew(const double* v, double* K) { out3(skew_sym_mat(v3(v)), K); }
void ref_state_init(double* pod) { StatesGroup s; to_pod(s, pod); }
void ref_state_boxplus(double* pod, const double* d24) {  // operator+= (src/laserMapping.cpp:1085 uses `state += solution`)
  StatesGroup s = from_pod(pod);
  Matrix<double, DIM_STATE, 1> d;
  for (int i = 0; i < DIM_STATE; i++) d(i, 0) = d24[i];
  s += d;
  to_pod(s, pod);
}
void ref_state_plus(const double* pod, const double* d24, double* out_pod) {  // operator+
  StatesGroup s = from_pod(pod);
  Matrix<double, DIM_STATE, 1> d;
  for (int i = 0; i < DIM_STATE; i++) d(i, 0) = d24[i];
  StatesGroup r = s + d;
  to_pod(r, out_pod);
}
void ref_state_boxminus(const double* a, const double* b, double* out24) {  // a - b
  StatesGroup sa = from_pod(a), sb = from_pod(b);
  Matrix<double, DIM_STATE, 1> d = sa - sb;
  for (int i = 0; i < DIM_STATE; i++) out24[i] = d(i, 0);
}
void ref_set_pose6d(double t, const double* a, const double* g, const double* v, const double* p, const double* R, double* out22) {
  Pose6D k = set_pose6d(t, v3(a), v3(g), v3(v), v3(p), m3(R));
  out22[0] = k.offset_time;
  for (int i = 0; i < 3; i++) { out22[1 + i] = k.acc[i]; out22[4 + i] = k.gyr[i]; out22[7 + i] = k.vel[i]; out22[10 + i] = k.pos[i]; }
  for (int i = 0; i < 9; i++) out22[13 + i] = k.rot[i];
}
double ref_rad2deg(double r) { return rad2deg(r); }
}
