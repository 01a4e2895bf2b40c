// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.hpp header).
//
// Ingest: wire-format decode, blind / NaN / ring / decimation filters, per-point time synthesis and sub-frame
// cutting — a CPU restatement of
//   Preprocess::process_cut_frame_livox   reference src/preprocess.cpp:50-113
//   Preprocess::process_cut_frame_pcl2    reference src/preprocess.cpp:115-335
//   Preprocess::process (PointCloud2: oust_handler / velodyne_handler / l515_handler; CustomMsg: avia_handler), feature extraction
//       disabled                            reference src/preprocess.cpp:337-713 - the callbacks' branch for initialization/cut_frame:
//       false (src/laserMapping.cpp:337-342, :374-379): selected here by IngestOpts::cut_frame_num == 0
// for the point layouts the reference registers (src/preprocess.h:35-116):
//   velodyne_ros::Point  x y z f32, intensity f32, time f32 [s], ring u16
//   ouster_ros::Point    x y z f32, intensity f32, t u32 [ns], ring u8
//   pandar_ros::Point    x y z f32, intensity f32, timestamp f64 [s], ring u16
//   robosense_ros::Point x y z f32, intensity u8, ring u16, timestamp f64 [s]
//   livox CustomPoint    offset_time u32 [ns], x y z f32, reflectivity u8, tag u8, line u8
// pcl::fromROSMsg (third-party, not vendored) maps message fields onto those structs by name; here the caller passes the
// byte offsets of the fields inside one point record, which is the information fromROSMsg derives from msg->fields.
//
// Deviation (documented, unavoidable): the reference orders the kept points with std::sort on `curvature`
// (src/preprocess.cpp:86,296), which is not stable — the order of points with EQUAL time stamps (e.g. the 128 returns of
// one Ouster column) is implementation-defined there.  Here ties keep their input order (std::stable_sort).
// PINNED: tests/test_oracle_ingest.py runs this restatement against the UNMODIFIED reference src/preprocess.cpp (built
// out-of-tree by oracle/Makefile into oracle/_ref/libref_preprocess.so) and against tests/golden/ingest/reference_frames.npz
// (outputs of that library, committed).  Only the tie order is unpinned; every count, boundary and time value is
// independent of it except which of several equal-time points sits at a sub-frame boundary.
//
// Quirks reproduced: the cut loop and the Livox decode loop start at index 1; the boundary test is
// `valid_num == int((cut_num + 1) * size / required) - 1` in unsigned arithmetic; the first 5 (Livox) / 20
// (PointCloud2) messages are not cut; points after the last boundary that is hit are dropped; a boundary that can never
// be hit (tiny clouds) stops all further cutting.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "orc_scan.hpp"

namespace orc {

enum LidType { AVIA = 1, VELO = 2, OUSTER = 3, L515 = 4, PANDAR = 5, ROBOSENSE = 6 };  // include/common_lib.h:55

struct Pc2Fields {  // byte offsets inside one point record of sensor_msgs/PointCloud2::data
  int point_step;
  int x, y, z, intensity, time, ring;
};
struct LivoxFields {  // byte offsets inside one livox_ros_driver/CustomPoint record as the caller stores it
  int point_step;
  int offset_time, x, y, z, reflectivity, tag, line;
};
struct IngestOpts {
  int lidar_type;        // LidType
  int n_scans;           // N_SCANS  (preprocess/scan_line)
  int point_filter_num;  // point_filter_num
  double blind;          // preprocess/blind
  double stamp_s;        // msg->header.stamp.toSec()
  int cut_frame_num;     // required_frame_num (initialization/cut_frame_num)
  int scan_count;        // scan_count of the caller (laserMapping.cpp:312,349)
};
struct Frame {
  double begin_time_ms;  // value pushed to time_lidar (the caller divides by 1000, laserMapping.cpp:334,370)
  std::vector<P4> pts;   // x, y, z, curvature [ms, relative to begin_time]
};

template <class T>
inline T rd(const uint8_t* p) {
  T v;
  std::memcpy(&v, p, sizeof(T));
  return v;
}

// the common tail of both functions: time sort + cutting (src/preprocess.cpp:86-112 == :296-334)
inline void cut_frames(std::vector<P4>& pl_surf, const IngestOpts& o, int uncut_below, std::vector<Frame>& out) {
  std::stable_sort(pl_surf.begin(), pl_surf.end(), [](const P4& a, const P4& b) { return a.t < b.t; });
  double last_frame_end_time = o.stamp_s * 1000;
  unsigned valid_num = 0, cut_num = 0;
  const unsigned valid_pcl_size = (unsigned)pl_surf.size();
  int required_cut_num = o.cut_frame_num;
  if (o.scan_count < uncut_below) required_cut_num = 1;
  std::vector<P4> pcl_cut;
  for (unsigned i = 1; i < valid_pcl_size; i++) {
    valid_num++;
    pl_surf[i].t += o.stamp_s * 1000 - last_frame_end_time;  // float += double
    pcl_cut.push_back(pl_surf[i]);
    if ((int)valid_num == (int((cut_num + 1) * valid_pcl_size / (unsigned)required_cut_num) - 1)) {
      cut_num++;
      Frame f;
      f.begin_time_ms = last_frame_end_time;
      f.pts = pcl_cut;
      out.push_back(std::move(f));
      last_frame_end_time += pl_surf[i].t;
      pcl_cut.clear();
    }
  }
}

// Preprocess::process(PointCloud2) with feature_enabled == false — src/preprocess.cpp:337-354 -> oust_handler :472-565 (the else
// branch :544-564), velodyne_handler :567-702 (:660-701), l515_handler :444-470.  No time sort, no cut, no point dropped at index
// 0: the cloud as the handler leaves it in pl_surf, input order; the caller stamps it with header.stamp (laserMapping.cpp:340,377).
// Any other lidar_type prints "Error LiDAR Type" and hands back whatever pl_surf held before: -1 here.
inline int ingest_pcl2_whole(const uint8_t* data, int plsize, const Pc2Fields& f, const IngestOpts& o, std::vector<Frame>& out) {
  constexpr int MAX_LINE_NUM = 128;
  if (o.lidar_type != VELO && o.lidar_type != OUSTER && o.lidar_type != L515) return -1;
  Frame fr;
  fr.begin_time_ms = o.stamp_s * 1000;
  bool given_offset_time = true;
  bool is_first[MAX_LINE_NUM];
  double yaw_fp[MAX_LINE_NUM] = {0};
  const double omega_l = 3.61;
  float time_last[MAX_LINE_NUM] = {0.0f};
  if (o.lidar_type == VELO && plsize > 0) {
    given_offset_time = (double)rd<float>(data + (size_t)(plsize - 1) * f.point_step + f.time) > 0;  // (:586-592)
    if (!given_offset_time) std::memset(is_first, true, sizeof(is_first));
  }
  for (int i = 0; i < plsize; i++) {
    const uint8_t* p = data + (size_t)i * f.point_step;
    P4 a;
    a.x = rd<float>(p + f.x);
    a.y = rd<float>(p + f.y);
    a.z = rd<float>(p + f.z);
    if (o.lidar_type == L515) {  // (:455-469): decimation first, no NaN test (a NaN range is not < blind^2), no ring, time 0
      if (i % o.point_filter_num != 0) continue;
      const double range = a.x * a.x + a.y * a.y + a.z * a.z;
      if (range < o.blind * o.blind) continue;
      a.t = 0.0f;
      fr.pts.push_back(a);
    } else if (o.lidar_type == OUSTER) {  // (:545-563)
      if (i % o.point_filter_num != 0) continue;
      a.t = (float)(rd<uint32_t>(p + f.time) / 1e6);
      const double dist = a.x * a.x + a.y * a.y + a.z * a.z;
      if (dist < o.blind * o.blind || std::isnan(a.x) || std::isnan(a.y) || std::isnan(a.z)) continue;
      if (rd<uint8_t>(p + f.ring) < o.n_scans) fr.pts.push_back(a);
    } else {  // VELO (:661-700): no ring filter in this branch; the decimation comes last
      a.t = (float)(rd<float>(p + f.time) * 1000.0);
      const double dist = a.x * a.x + a.y * a.y + a.z * a.z;
      if (dist < o.blind * o.blind || std::isnan(a.x) || std::isnan(a.y) || std::isnan(a.z)) continue;
      if (!given_offset_time) {
        const int layer = rd<uint16_t>(p + f.ring);
        if (layer < 0 || layer >= MAX_LINE_NUM) continue;  // the reference indexes out of bounds here; such points are dropped
        const double yaw_angle = std::atan2(a.y, a.x) * 57.2957;
        if (is_first[layer]) {
          yaw_fp[layer] = yaw_angle;
          is_first[layer] = false;
          time_last[layer] = 0.0f;
          continue;
        }
        if (yaw_angle <= yaw_fp[layer]) a.t = (float)((yaw_fp[layer] - yaw_angle) / omega_l);
        else a.t = (float)((yaw_fp[layer] - yaw_angle + 360.0) / omega_l);
        if (a.t < time_last[layer]) a.t = (float)(a.t + 360.0 / omega_l);
        time_last[layer] = a.t;
      }
      if (i % o.point_filter_num == 0) fr.pts.push_back(a);
    }
  }
  out.push_back(std::move(fr));
  return 0;
}

// process_cut_frame_pcl2 — src/preprocess.cpp:115-335
inline int ingest_pcl2(const uint8_t* data, int plsize, const Pc2Fields& f, const IngestOpts& o, std::vector<Frame>& out) {
  if (o.cut_frame_num == 0) return ingest_pcl2_whole(data, plsize, f, o, out);
  std::vector<P4> pl_surf;
  pl_surf.reserve(plsize);
  constexpr int MAX_LINE_NUM = 128;
  const bool synth = (o.lidar_type == VELO || o.lidar_type == ROBOSENSE);
  bool given_offset_time = true;
  bool is_first[MAX_LINE_NUM];
  double yaw_fp[MAX_LINE_NUM] = {0};
  const double omega_l = 3.61;  // deg / ms
  float yaw_last[MAX_LINE_NUM] = {0.0f};
  float time_last[MAX_LINE_NUM] = {0.0f};
  if (o.lidar_type != VELO && o.lidar_type != OUSTER && o.lidar_type != PANDAR && o.lidar_type != ROBOSENSE) return -1;
  if (synth && plsize > 0) {
    const uint8_t* last = data + (size_t)(plsize - 1) * f.point_step;
    const double tl = o.lidar_type == VELO ? (double)rd<float>(last + f.time) : rd<double>(last + f.time);
    if (tl > 0) {
      given_offset_time = true;
    } else {
      given_offset_time = false;
      std::memset(is_first, true, sizeof(is_first));
    }
  }
  const double ts0 = (o.lidar_type == PANDAR && plsize > 0) ? rd<double>(data + f.time) : 0.0;
  for (int i = 0; i < plsize; i++) {
    const uint8_t* p = data + (size_t)i * f.point_step;
    P4 a;
    a.x = rd<float>(p + f.x);
    a.y = rd<float>(p + f.y);
    a.z = rd<float>(p + f.z);
    int ring;
    switch (o.lidar_type) {
      case VELO:
        a.t = (float)(rd<float>(p + f.time) * 1000.0);  // s -> ms (:153)
        ring = rd<uint16_t>(p + f.ring);
        break;
      case OUSTER:
        a.t = (float)(rd<uint32_t>(p + f.time) / 1e6);  // ns -> ms (:202)
        ring = rd<uint8_t>(p + f.ring);
        break;
      case PANDAR:
        a.t = (float)((rd<double>(p + f.time) - ts0) * 1000);  // (:228)
        ring = rd<uint16_t>(p + f.ring);
        break;
      default:  // ROBOSENSE
        a.t = (float)((rd<double>(p + f.time) - o.stamp_s + 0.1) * 1000.0);  // (:265)
        ring = rd<uint16_t>(p + f.ring);
        break;
    }
    const double dist = a.x * a.x + a.y * a.y + a.z * a.z;  // float arithmetic, widened on assignment
    if (dist < o.blind * o.blind || std::isnan(a.x) || std::isnan(a.y) || std::isnan(a.z)) continue;
    if (synth && !given_offset_time) {
      const int layer = ring;
      if (layer < 0 || layer >= MAX_LINE_NUM) continue;  // the reference indexes out of bounds here; such points are dropped
      const double yaw_angle = std::atan2(a.y, a.x) * 57.2957;
      if (is_first[layer]) {
        yaw_fp[layer] = yaw_angle;
        is_first[layer] = false;
        a.t = 0.0f;
        yaw_last[layer] = (float)yaw_angle;
        time_last[layer] = a.t;
        continue;
      }
      if (yaw_angle <= yaw_fp[layer]) {
        a.t = (float)((yaw_fp[layer] - yaw_angle) / omega_l);
      } else {
        a.t = (float)((yaw_fp[layer] - yaw_angle + 360.0) / omega_l);
      }
      if (a.t < time_last[layer]) a.t = (float)(a.t + 360.0 / omega_l);
      yaw_last[layer] = (float)yaw_angle;
      time_last[layer] = a.t;
    }
    if (i % o.point_filter_num == 0 && ring < o.n_scans) pl_surf.push_back(a);
  }
  cut_frames(pl_surf, o, 20, out);
  return 0;
}

// process_cut_frame_livox — src/preprocess.cpp:50-113; cut_frame_num == 0: Preprocess::process(CustomMsg) = avia_handler with
// feature_enabled == false (:355-443, the else branch :419-442) - the same per-point loop, then neither the time sort nor the cut:
// pl_surf in input order, its first point included.
inline int ingest_livox(const uint8_t* data, int plsize, const LivoxFields& f, const IngestOpts& o, std::vector<Frame>& out) {
  std::vector<P4> pl_surf;
  pl_surf.reserve(plsize);
  std::vector<P4> pl_full((size_t)std::max(plsize, 0), P4{0, 0, 0, 0});  // pl_full.resize(plsize): zero-initialised points
  int valid_point_num = 0;
  for (int i = 1; i < plsize; i++) {
    const uint8_t* p = data + (size_t)i * f.point_step;
    const int line = rd<uint8_t>(p + f.line), tag = rd<uint8_t>(p + f.tag);
    if ((line < o.n_scans) && ((tag & 0x30) == 0x10 || (tag & 0x30) == 0x00)) {
      valid_point_num++;
      if (valid_point_num % o.point_filter_num == 0) {
        pl_full[i].x = rd<float>(p + f.x);
        pl_full[i].y = rd<float>(p + f.y);
        pl_full[i].z = rd<float>(p + f.z);
        pl_full[i].t = rd<uint32_t>(p + f.offset_time) / float(1000000);  // ns -> ms, float division (:69)
        const double dist = pl_full[i].x * pl_full[i].x + pl_full[i].y * pl_full[i].y + pl_full[i].z * pl_full[i].z;
        if (dist < o.blind * o.blind) continue;
        if ((std::abs(pl_full[i].x - pl_full[i - 1].x) > 1e-7) || (std::abs(pl_full[i].y - pl_full[i - 1].y) > 1e-7) ||
            (std::abs(pl_full[i].z - pl_full[i - 1].z) > 1e-7)) {
          pl_surf.push_back(pl_full[i]);
        }
      }
    }
  }
  if (o.cut_frame_num == 0) {
    Frame fr;
    fr.begin_time_ms = o.stamp_s * 1000;
    fr.pts = pl_surf;
    out.push_back(std::move(fr));
    return 0;
  }
  cut_frames(pl_surf, o, 5, out);
  return 0;
}

}  // namespace orc
