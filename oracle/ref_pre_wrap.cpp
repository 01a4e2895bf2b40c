// ORACLE — TEST INFRASTRUCTURE ONLY.  extern "C" wrapper around the UNMODIFIED reference Preprocess class
// (/root/reference/src/preprocess.{h,cpp}, compiled where it lies against oracle/ref_shim_pre): builds the message
// structs from raw bytes + field offsets and calls Preprocess::process_cut_frame_pcl2 / process_cut_frame_livox - or, with
// cut_frame_num == 0, Preprocess::process (the callbacks' branch for initialization/cut_frame: false).
// Used by tests/golden/make_ingest_fixture.py to pin oracle/orc_ingest.hpp against the reference's own code.
#include "preprocess.h"

namespace {
int flatten(deque<PointCloudXYZI::Ptr>& pcl_out, deque<double>& time_lidar, float* out4, int cap_pts, double* begin_ms, int* offsets,
            int* counts, int cap_frames) {
  if ((int)pcl_out.size() > cap_frames) return -2;
  int off = 0;
  for (size_t k = 0; k < pcl_out.size(); k++) {
    const PointCloudXYZI& c = *pcl_out[k];
    if (off + (int)c.size() > cap_pts) return -2;
    begin_ms[k] = time_lidar[k];
    offsets[k] = off;
    counts[k] = (int)c.size();
    for (size_t i = 0; i < c.size(); i++) {
      out4[4 * (off + i) + 0] = c.points[i].x;
      out4[4 * (off + i) + 1] = c.points[i].y;
      out4[4 * (off + i) + 2] = c.points[i].z;
      out4[4 * (off + i) + 3] = c.points[i].curvature;
    }
    off += (int)c.size();
  }
  return (int)pcl_out.size();
}
}  // namespace

extern "C" {

int ref_ingest_pcl2(const unsigned char* data, int n, const int* fields7, int lidar_type, int n_scans, int point_filter_num,
                    double blind, double stamp_s, int cut_frame_num, int scan_count, float* out4, int cap_pts, double* begin_ms,
                    int* offsets, int* counts, int cap_frames) {
  auto msg = std::make_shared<sensor_msgs::PointCloud2>();
  msg->header.stamp.t = stamp_s;
  msg->point_step = fields7[0];
  msg->width = n;
  msg->row_step = n * fields7[0];
  msg->data.assign(data, data + (size_t)n * fields7[0]);
  const char* tname = lidar_type == VELO ? "time" : (lidar_type == OUSTER ? "t" : "timestamp");
  const char* names[6] = {"x", "y", "z", "intensity", tname, "ring"};
  for (int k = 0; k < 6; k++) {
    sensor_msgs::PointField f;
    f.name = names[k];
    f.offset = fields7[1 + k];
    msg->fields.push_back(f);
  }
  Preprocess pre;
  pre.set(false, lidar_type, blind, point_filter_num);
  pre.N_SCANS = n_scans;
  deque<PointCloudXYZI::Ptr> pcl_out;
  deque<double> time_lidar;
  if (cut_frame_num == 0) {  // src/laserMapping.cpp:337-342: p_pre->process(msg, ptr); time_buffer.push_back(msg->header.stamp.toSec())
    if (lidar_type != VELO && lidar_type != OUSTER && lidar_type != L515) return -1;  // "Error LiDAR Type": pl_surf is whatever it was
    PointCloudXYZI::Ptr ptr(new PointCloudXYZI());
    pre.process(msg, ptr);
    pcl_out.push_back(ptr);
    time_lidar.push_back(stamp_s * 1000);
  } else {
    pre.process_cut_frame_pcl2(msg, pcl_out, time_lidar, cut_frame_num, scan_count);
  }
  return flatten(pcl_out, time_lidar, out4, cap_pts, begin_ms, offsets, counts, cap_frames);
}

int ref_ingest_livox(const unsigned char* data, int n, const int* fields8, int n_scans, int point_filter_num, double blind,
                     double stamp_s, int cut_frame_num, int scan_count, float* out4, int cap_pts, double* begin_ms, int* offsets,
                     int* counts, int cap_frames) {
  auto msg = std::make_shared<livox_ros_driver::CustomMsg>();
  msg->header.stamp.t = stamp_s;
  msg->point_num = n;
  msg->points.resize(n);
  for (int i = 0; i < n; i++) {
    const unsigned char* p = data + (size_t)i * fields8[0];
    livox_ros_driver::CustomPoint& q = msg->points[i];
    std::memcpy(&q.offset_time, p + fields8[1], 4);
    std::memcpy(&q.x, p + fields8[2], 4);
    std::memcpy(&q.y, p + fields8[3], 4);
    std::memcpy(&q.z, p + fields8[4], 4);
    q.reflectivity = p[fields8[5]];
    q.tag = p[fields8[6]];
    q.line = p[fields8[7]];
  }
  Preprocess pre;
  pre.set(false, AVIA, blind, point_filter_num);
  pre.N_SCANS = n_scans;
  deque<PointCloudXYZI::Ptr> pcl_out;
  deque<double> time_lidar;
  if (cut_frame_num == 0) {  // src/laserMapping.cpp:374-379
    PointCloudXYZI::Ptr ptr(new PointCloudXYZI());
    pre.process(msg, ptr);
    pcl_out.push_back(ptr);
    time_lidar.push_back(stamp_s * 1000);
  } else {
    pre.process_cut_frame_livox(msg, pcl_out, time_lidar, cut_frame_num, scan_count);
  }
  return flatten(pcl_out, time_lidar, out4, cap_pts, begin_ms, offsets, counts, cap_frames);
}

}  // extern "C"
