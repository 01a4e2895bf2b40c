// Build shim (OURS): pcl::PointXYZINormal / PointXYZRGB layouts and the registration macros src/preprocess.h uses.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#define EIGEN_ALIGN16 alignas(16)
#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW
#define PCL_ADD_POINT4D \
  union {               \
    float data[4];      \
    struct {            \
      float x, y, z;    \
    };                  \
  };

namespace pcl {
struct alignas(16) PointXYZINormal {
  float x = 0.f, y = 0.f, z = 0.f, data_pad = 1.f;
  float normal_x = 0.f, normal_y = 0.f, normal_z = 0.f, normal_pad = 0.f;
  float intensity = 0.f, curvature = 0.f, tail_pad0 = 0.f, tail_pad1 = 0.f;
};
struct alignas(16) PointXYZRGB {
  float x = 0.f, y = 0.f, z = 0.f, data_pad = 1.f;
  std::uint8_t b = 0, g = 0, r = 0, a = 255;
  float pad[3] = {0, 0, 0};
};
// one registered field of a point struct: name in the message, where it lives in the struct, how many bytes
struct ShimField {
  std::string name;
  std::size_t offset, size;
};
template <class P>
struct ShimFields;  // specialised by POINT_CLOUD_REGISTER_POINT_STRUCT
}  // namespace pcl

// (type, member, tag)(type, member, tag)... -> one push_back per triple (the classic alternating-macro sequence walk)
#define PCL_SHIM_REG_A(t, m, n) v.push_back({#n, offsetof(P_, m), sizeof(t)}); PCL_SHIM_REG_B
#define PCL_SHIM_REG_B(t, m, n) v.push_back({#n, offsetof(P_, m), sizeof(t)}); PCL_SHIM_REG_A
#define PCL_SHIM_REG_A_END
#define PCL_SHIM_REG_B_END
#define PCL_SHIM_CAT_(a, b) a##b
#define PCL_SHIM_CAT(a, b) PCL_SHIM_CAT_(a, b)
#define POINT_CLOUD_REGISTER_POINT_STRUCT(name, seq)          \
  namespace pcl {                                             \
  template <>                                                 \
  struct ShimFields<name> {                                   \
    static std::vector<ShimField> get() {                     \
      typedef name P_;                                        \
      std::vector<ShimField> v;                               \
      PCL_SHIM_CAT(PCL_SHIM_REG_A seq, _END)                  \
      return v;                                               \
    }                                                         \
  };                                                          \
  }
