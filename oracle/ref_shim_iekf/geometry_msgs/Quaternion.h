// Build shim (OURS): the message struct the update's (unused) pose publication fills (src/laserMapping.cpp:144, 1118-1126).
#pragma once
namespace geometry_msgs {
struct Quaternion { double x = 0, y = 0, z = 0, w = 1; };
}  // namespace geometry_msgs
