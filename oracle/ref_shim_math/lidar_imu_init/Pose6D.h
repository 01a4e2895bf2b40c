// Build shim (OURS): msg/Pose6D.msg as the struct rosmsg generates the fields of.
#pragma once
namespace lidar_imu_init {
struct Pose6D {
  double offset_time = 0;
  double acc[3] = {0, 0, 0}, gyr[3] = {0, 0, 0}, vel[3] = {0, 0, 0}, pos[3] = {0, 0, 0};
  double rot[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
};
}  // namespace lidar_imu_init
