// Build shim (OURS): livox_ros_driver/CustomMsg + CustomPoint (v2.6.0 message definitions) as plain structs.
#pragma once
#include <cstdint>
#include <memory>
#include <vector>
#include <ros/ros.h>
namespace livox_ros_driver {
struct CustomPoint {
  std::uint32_t offset_time;
  float x, y, z;
  std::uint8_t reflectivity, tag, line;
};
struct CustomMsg {
  typedef std::shared_ptr<const CustomMsg> ConstPtr;
  std_msgs::Header header;
  std::uint64_t timebase = 0;
  std::uint32_t point_num = 0;
  std::uint8_t lidar_id = 0;
  std::uint8_t rsvd[3] = {0, 0, 0};
  std::vector<CustomPoint> points;
};
}  // namespace livox_ros_driver
