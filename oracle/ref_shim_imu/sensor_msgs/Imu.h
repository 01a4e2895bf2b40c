// Build shim (OURS): the fields of sensor_msgs::Imu that src/IMU_Processing.hpp reads (header stamp, angular velocity, linear
// acceleration) as a plain struct; boost::shared_ptr -> std::shared_ptr.  Test infrastructure only.
#pragma once
#include <memory>
namespace sensor_msgs {
struct ImuStamp {
  double sec = 0;
  double toSec() const { return sec; }
};
struct ImuHeader {
  ImuStamp stamp;
};
struct ImuVec3 {
  double x = 0, y = 0, z = 0;
};
struct Imu {
  typedef std::shared_ptr<const Imu> ConstPtr;
  typedef std::shared_ptr<Imu> Ptr;
  ImuHeader header;
  ImuVec3 angular_velocity, linear_acceleration;
};
typedef Imu::ConstPtr ImuConstPtr;
}  // namespace sensor_msgs
