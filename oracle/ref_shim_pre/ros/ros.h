// Build shim (OURS): the two ROS types src/preprocess.{h,cpp} name.
#pragma once
#include <string>
namespace ros {
struct Time {
  double t = 0;  // header.stamp.toSec() is the only accessor the ingest path uses
  double toSec() const { return t; }
};
struct Publisher {};
}  // namespace ros
namespace std_msgs {
struct Header {
  unsigned seq = 0;
  ros::Time stamp;
  std::string frame_id;
};
}  // namespace std_msgs
