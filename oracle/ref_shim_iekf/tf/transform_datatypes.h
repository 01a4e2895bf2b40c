// Build shim (OURS): tf::createQuaternionMsgFromRollPitchYaw (fixed-axis roll-pitch-yaw -> quaternion), called by the reference's
// covariance block for a message nothing in the sliced path reads (src/laserMapping.cpp:1118-1126).
#pragma once
#include <cmath>

#include <geometry_msgs/Quaternion.h>
namespace tf {
inline geometry_msgs::Quaternion createQuaternionMsgFromRollPitchYaw(double roll, double pitch, double yaw) {
  const double cr = std::cos(0.5 * roll), sr = std::sin(0.5 * roll), cp = std::cos(0.5 * pitch), sp = std::sin(0.5 * pitch);
  const double cy = std::cos(0.5 * yaw), sy = std::sin(0.5 * yaw);
  geometry_msgs::Quaternion q;
  q.x = sr * cp * cy - cr * sp * sy;
  q.y = cr * sp * cy + sr * cp * sy;
  q.z = cr * cp * sy - sr * sp * cy;
  q.w = cr * cp * cy + sr * sp * sy;
  return q;
}
}  // namespace tf
