// Build shim (OURS): pcl::fromROSMsg / toROSMsg.  fromROSMsg copies, for every field the point struct registered, the
// bytes of the message field with the same name (PCL's field mapping by name; datatypes are taken to match, as they
// do for the drivers the reference supports).  A field missing from the message keeps its default.
#pragma once
#include <cstring>
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#include <sensor_msgs/PointCloud2.h>
namespace pcl {
template <class P>
void fromROSMsg(const sensor_msgs::PointCloud2& msg, PointCloud<P>& cloud) {
  const std::vector<ShimField> reg = ShimFields<P>::get();
  const std::size_t n = msg.point_step ? msg.data.size() / msg.point_step : 0;
  cloud.points.assign(n, P());
  cloud.width = (std::uint32_t)n;
  cloud.height = 1;
  cloud.is_dense = msg.is_dense;
  for (const ShimField& f : reg)
    for (const sensor_msgs::PointField& mf : msg.fields)
      if (mf.name == f.name)
        for (std::size_t i = 0; i < n; i++)
          std::memcpy(reinterpret_cast<char*>(&cloud.points[i]) + f.offset, msg.data.data() + i * msg.point_step + mf.offset, f.size);
}
template <class P>
void toROSMsg(const PointCloud<P>&, sensor_msgs::PointCloud2&) {}
}  // namespace pcl
// the reference registers pcl::PointXYZRGB through PCL itself; src/preprocess.cpp reads an L515 cloud into it
POINT_CLOUD_REGISTER_POINT_STRUCT(pcl::PointXYZRGB, (float, x, x)(float, y, y)(float, z, z))
POINT_CLOUD_REGISTER_POINT_STRUCT(pcl::PointXYZINormal, (float, x, x)(float, y, y)(float, z, z)(float, intensity, intensity))
