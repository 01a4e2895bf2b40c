// Internal device-side declarations of libliinit_hip (gfx950 only).  Not part of the C-ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lii {

constexpr int kMatch = 5;          // NUM_MATCH_POINTS — reference include/common_lib.h:28
constexpr int kBlock = 256;        // 4 wavefronts of 64
constexpr int kNormalEq = 91;      // 78 + 12 + 1
constexpr int kCoarseShift = 3;    // coarse occupancy cell = 8 x 8 x 8 fine cells
constexpr unsigned long long kEmptyKey = ~0ull;
constexpr int kCellBias = 1 << 20;

// Pose the per-point kernels need: state.rot_end, pos_end, offset_R_L_I, offset_T_L_I (row-major).
struct PoseArg {
  double R[9];
  double p[3];
  double RLI[9];
  double TLI[3];
};

struct __attribute__((aligned(16))) CellEntry {
  unsigned long long key;  // packed (cz, cy, cx); kEmptyKey = free slot
  unsigned int start;      // first index into the cell-sorted point array
  unsigned int end;        // one past the last
};

// Device view of the local-map k-NN index: points sorted by fine-cell key + two open-addressing tables.
struct GridView {
  const float4* pts;  // xyz + w = bit-cast insertion id
  const CellEntry* fine;
  const unsigned long long* coarse;
  unsigned int fine_mask;
  unsigned int coarse_mask;
  int n_pts;
  float cs;
  float inv_cs;
  float max_d2;
};

struct RegistrationBuffers {
  const float4* body;   // down-sampled LiDAR-frame points (x,y,z,t)
  float4* world;        // world coordinates of the last pass (x,y,z,-)
  float4* nbr;          // SoA: nbr[k * cap + i], k < 5 (xyz of neighbour k, w = d2)
  int* nbr_count;       // neighbours found (0..5)
  double* plane;        // 4 doubles per point: n̂, d  (pabcd)
  unsigned char* selected;
  double* partials;     // per-block 91 doubles
  int n;
  int cap;
};

}  // namespace lii
