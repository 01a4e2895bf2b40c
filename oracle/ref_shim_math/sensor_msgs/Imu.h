// Build shim (OURS): only the pointer typedef common_lib.h's MeasureGroup names.
#pragma once
#include <memory>
namespace sensor_msgs {
struct Imu { typedef std::shared_ptr<const Imu> ConstPtr; };
}  // namespace sensor_msgs
