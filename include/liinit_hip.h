/* liinit_hip.h — C-ABI of the MI355X-native LI-Init hot path (libliinit_hip.so).
 *
 * The reference (hku-mars/LiDAR_IMU_Init) has NO plugin / FFI seam: its hot path is inline C++ over
 * globals in one ROS executable.  This header therefore DEFINES the boundary a
 * maintainer would bind to; every entry point names the reference code it replaces (file:line relative
 * to the reference root).  Conventions:
 *   - plain C, opaque handle, `int` status (0 = LII_OK, negative = error; lii_strerror / lii_last_error);
 *   - caller-owned host buffers, library-owned device buffers; no exceptions cross the boundary;
 *   - one handle is used from one host thread; all device work of a handle runs on ONE HIP stream;
 *   - the library FAILS (LII_ERR_NO_DEVICE) when no gfx950 device is usable — there is no CPU fallback.
 */
#ifndef LIINIT_HIP_H
#define LIINIT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LII_ABI_VERSION 9 /* 2: lii_scan_job::scan_dev / n_scan_dev, lii_comm_init_ex, lii_comm_transport
                             3: lii_params_*, lii_comm_set_partition (library-side split of the down-sampled cloud)
                             4: lii_scan_upload_next / lii_scan_advance (the next scan travels while the current one registers)
                             5: LII_COMM_MAILBOX = the peer-mapped HBM mailbox (HIP IPC), LII_COMM_MAILBOX_HOST, lii_comm_rccl_ranks
                             6: lii_scan_job::scan_sorted (struct_size 56; a job of size 48 - ABI 5 - is still accepted), lii_last_kernel_profile,
                                lii_comm_describe, lii_comm_set_partition(h, 2) (split by voxel), lii_scan_job::map_update (the reserved field)
                             7: lii_ingest_opts::cut_frame_num = 0 (the whole message as one frame: Preprocess::process), lii_last_solve_info,
                                lii_selftest_list_exchange
                             8: lii_last_unfinished_queries, lii_scan_job::next_scan_dev / next_n_scan (struct_size 72: the pre-armed prologue)
                             9: lii_ingest_pcl2_begin / lii_ingest_livox_begin / lii_ingest_end (driver messages queue on the device: message k + 1 is
                                transferred and decoded while the sub-frames of message k are registered), lii_scan_job::while_waiting (struct_size 88) */

enum lii_status {
  LII_OK = 0,
  LII_ERR_INVALID = -1,   /* bad argument / NULL / size */
  LII_ERR_NO_DEVICE = -2, /* no HIP device (the library never falls back to the CPU) */
  LII_ERR_HIP = -3,       /* a HIP runtime call failed; see lii_last_error */
  LII_ERR_CAPACITY = -4,  /* more points than the handle was created for */
  LII_ERR_STATE = -5,     /* call order violated (e.g. registration without a map) */
  LII_ERR_COMM = -6       /* RCCL failure */
};

typedef struct lii_context* lii_handle;

/* Replaces the file-scope configuration of the reference: NUM_MATCH_POINTS 5 (include/common_lib.h:28),
 * Nearest_Search max_dist 5 (src/laserMapping.cpp:980), esti_plane threshold 0.1 (:997),
 * LASER_POINT_COV 0.001 (:59), the fixed 100000-point arrays (:108-109,117-119 — lifted to a capacity). */
typedef struct lii_config {
  int32_t struct_size;       /* = sizeof(lii_config) */
  int32_t device;            /* HIP device ordinal */
  int32_t max_scan_points;   /* capacity of one (sub-)scan, N */
  int32_t max_map_points;    /* capacity of the local map, M */
  float map_cell_size;       /* edge of the device k-NN grid cell [m]; <= 0: 3 x map_downsample_size */
  float map_downsample_size; /* ikd-Tree downsample box = mapping/filter_size_map (set_downsample_param) */
  float max_match_dist2;     /* accept neighbours with d^2 <= this (5.0) */
  float reserved0;
  double plane_threshold;     /* 0.1 */
  double laser_point_cov_inv; /* 1000 */
} lii_config;

/* IMU pose-table record: msg/Pose6D.msg:1-7 as filled by set_pose6d (include/common_lib.h:183-199). */
typedef struct lii_pose6d {
  double offset_time; /* [s] from scan start */
  double acc[3], gyr[3], vel[3], pos[3];
  double rot[9]; /* row-major */
} lii_pose6d;

/* StatesGroup (include/common_lib.h:68-169) as POD; rotation matrices row-major; cov 24x24 row-major.
 * State order: theta, p, theta_LI, T_LI, v, b_g, b_a, g (Appendix B of SURVEY.md). */
typedef struct lii_state {
  double rot_end[9];
  double pos_end[3];
  double offset_R_L_I[9];
  double offset_T_L_I[3];
  double vel_end[3];
  double bias_g[3];
  double bias_a[3];
  double gravity[3];
  double cov[24 * 24];
} lii_state;

/* Options of one scan registration — the loop constants of src/laserMapping.cpp:957-1134. */
typedef struct lii_iekf_opts {
  int32_t max_iterations; /* NUM_MAX_ITERATIONS (param max_iteration; launch files: 5) */
  int32_t imu_en;         /* 0: LiDAR-only odometry (H cols 6..11 = 0), 1: LIO with extrinsic in state */
} lii_iekf_opts;

typedef struct lii_iekf_report {
  int32_t iterations;    /* executed */
  int32_t searches;      /* k-NN passes executed (S) */
  int32_t effect_num;    /* effect_feat_num of the last iteration */
  int32_t converged;     /* flg_EKF_converged of the last iteration */
  double normal_eq[91];  /* last iteration: upper 78 of H^T R^-1 H, 12 of H^T R^-1 z, count */
} lii_iekf_report;

/* ---------------------------------------------------------------- lifecycle */
int lii_abi_version(void);
const char* lii_strerror(int status);
const char* lii_last_error(lii_handle h);
int lii_device_count(int* count);
/* Replaces the global KD_TREE / buffers of src/laserMapping.cpp:102-150. */
int lii_create(const lii_config* cfg, lii_handle* out);
int lii_destroy(lii_handle h);
int lii_synchronize(lii_handle h);

/* ---------------------------------------------------------------- local map (device mirror of the ikd-Tree)
 * lii_map_build      <- ikdtree.Build(feats_down_world->points)           src/laserMapping.cpp:921-931
 * lii_map_add_points <- ikdtree.Add_Points(PointToAdd, true|false)        src/laserMapping.cpp:556-557,
 *                       with the per-voxel keep-closest-to-centre semantics of include/ikd-Tree/ikd_Tree.cpp:381-456
 * lii_map_size       <- ikdtree.validnum()                                src/laserMapping.cpp:933
 * Points are float xyz with `stride_bytes` between records (12 for packed xyz, 48 for PointXYZINormal).
 * The map is a point SET kept on the device and updated in place: lii_map_download returns it in no particular order;
 * an update is enqueued, not waited for (lii_map_size / lii_map_download / lii_map_add_points' counter read the device).
 * LII_ERR_CAPACITY: n_valid + batch would exceed lii_config::max_map_points - tested BEFORE the batch is applied, counting
 * every point of it; the map is left as it was.  The in-place layout keeps slack behind every cell (count + max(2, count / 4)
 * slots) and moves a cell that outgrows it to the tail of a point array of 3 * max_map_points + 65 536 slots: a SPARSE map (many
 * one-point cells: 3 slots each) close to max_map_points can need a rebuild on almost every update, and a batch of b inserts asks
 * for 8 b + 4 096 free tail slots after a rebuild - if even the rebuilt map cannot offer them the call fails with
 * LII_ERR_CAPACITY ("no room left behind the cells") although n_valid + b <= max_map_points.  Size max_map_points with ~25 %
 * headroom over the largest map expected.  An update that runs out of provisioned room while it executes loses nothing: the
 * inserts it could not place are parked and re-inserted by a rebuild before the next search or update uses the map. */
int lii_map_reset(lii_handle h);
int lii_map_build(lii_handle h, const void* xyz, int32_t n, int32_t stride_bytes);
int lii_map_add_points(lii_handle h, const void* xyz, int32_t n, int32_t stride_bytes, int32_t downsample_on,
                       int32_t* n_added);
/* lii_map_delete_boxes <- ikdtree.Delete_Point_Boxes(cub_needrm)  (include/ikd-Tree/ikd_Tree.cpp:500-516; the call the reference
 *                       prepares in lasermap_fov_segment, src/laserMapping.cpp:260-305, and never makes - quirk A6: its map only
 *                       grows).  boxes = n x 6 floats (min xyz, max xyz); a point goes when min <= p < max on every axis. */
int lii_map_delete_boxes(lii_handle h, const float* boxes, int32_t n_boxes, int32_t* n_deleted);
int lii_map_size(lii_handle h, int32_t* n_valid);
int lii_map_download(lii_handle h, float* xyz_out, int32_t capacity, int32_t* n);
/* Kept for callers of ABI 1: the device map is always current (updates are applied in place); waits for the stream. */
int lii_map_commit(lii_handle h);

/* ---------------------------------------------------------------- scan in / undistortion / down-sampling
 * lii_scan_upload: host AoS -> device float4 (x,y,z,t_ms).  For pcl::PointXYZINormal use stride 48,
 *                  time_offset_bytes 36 (`curvature`, include/common_lib.h:37).  Replaces the hand-over of
 *                  Measures.lidar to ImuProcess::Process (src/laserMapping.cpp:909).
 * lii_scan_set_device: same, from a device-resident float4 buffer (copied on the handle's stream by the kernel that also
 *                  reduces the scan's time extent; the caller's buffer is only read and free again once a later call on
 *                  the handle has returned).
 * ORDER OF THE POINTS.  The reference sorts every scan by time before it de-skews it (std::sort by curvature,
 *                  src/IMU_Processing.hpp:209, :287; its preprocess hands the scan over sorted as well,
 *                  src/preprocess.cpp:296-302) and everything downstream sees that order.  The library does NOT sort: the
 *                  de-skew itself does not need it (the time-earliest point is found by a reduction), but the voxel-grid
 *                  centroids are float sums in input order, so results are bit-identical to the reference's only for a
 *                  scan handed over in ascending time order (equal stamps in the reference's order) - which is what
 *                  lii_ingest_* / lii_frame_select deliver.  An unsorted scan is registered correctly up to the rounding of
 *                  those sums (~1e-6 m per centroid).  The voxel filter emits the down-sampled cloud in the order of the
 *                  voxels' first points; lii_scan_download(1 / 2) and lii_neighbors_download return the reference's order
 *                  (ascending PCL voxel index).
 * lii_undistort_imu <- back-propagation loop of ImuProcess::propagation_and_undist, src/IMU_Processing.hpp:390-414
 * lii_undistort_cv  <- CV de-skew of Forward_propagation_without_imu,            src/IMU_Processing.hpp:246-266
 * lii_downsample    <- downSizeFilterSurf.filter(*feats_down_body),              src/laserMapping.cpp:917-919
 *                      (n_down == NULL && filtered == NULL: fully asynchronous — the result size stays on the device)
 * lii_downsample_skip: use the (undistorted) scan as feats_down_body unchanged (PCL's overflow-identity path).
 * lii_scan_download: which = 0 undistorted scan, 1 down-sampled body points, 2 world points of the last iteration.
 * lii_scan_upload_next / lii_scan_advance: the scan queue of the reference (lidar_buffer, src/laserMapping.cpp:331-366 ->
 *                  sync_packages :432-480) one element deep, on the device.  lii_scan_upload_next starts the NEXT scan on its
 *                  way (same arguments as lii_scan_upload) into a second device buffer on the library's copy stream and returns
 *                  at once: the transfer overlaps the registration / map update of the current scan.  A source in pinned host
 *                  memory with stride 16 / time offset 12 is read by the copy engine directly and must stay untouched until
 *                  lii_scan_advance has returned; any other source is staged (copied before the call returns).
 *                  lii_scan_advance makes that scan the current one (as lii_scan_upload would have): waits for its transfer,
 *                  swaps the buffers.  LII_ERR_STATE without a pending lii_scan_upload_next. */
int lii_scan_upload(lii_handle h, const void* points, int32_t n, int32_t stride_bytes, int32_t time_offset_bytes);
int lii_scan_upload_next(lii_handle h, const void* points, int32_t n, int32_t stride_bytes, int32_t time_offset_bytes);
int lii_scan_advance(lii_handle h);
int lii_scan_set_device(lii_handle h, const void* dev_float4, int32_t n);
int lii_undistort_imu(lii_handle h, const lii_pose6d* poses, int32_t n_poses, const double end_R[9],
                      const double end_p[3], const double R_LI[9], const double T_LI[3]);
int lii_undistort_cv(lii_handle h, const double omega[3], const double vel[3], const double end_R[9]);
int lii_downsample(lii_handle h, float leaf, int32_t* n_down, int32_t* filtered);
int lii_downsample_skip(lii_handle h, int32_t* n_down);
int lii_scan_download(lii_handle h, int32_t which, float* out_float4, int32_t capacity, int32_t* n);

/* ---------------------------------------------------------------- ingest: driver message -> device-resident scan frames
 * lii_ingest_pcl2  <- Preprocess::process_cut_frame_pcl2  (src/preprocess.cpp:115-335; callers laserMapping.cpp:363-372)
 * lii_ingest_livox <- Preprocess::process_cut_frame_livox (src/preprocess.cpp:50-113;  callers laserMapping.cpp:326-336)
 * ... and, with lii_ingest_opts::cut_frame_num = 0, the callbacks' branch for `initialization/cut_frame: false`
 *                     (laserMapping.cpp:337-342, :374-379): Preprocess::process - oust_handler / velodyne_handler / l515_handler
 *                     (PointCloud2; any other lidar_type is "Error LiDAR Type" there, LII_ERR_INVALID here) and avia_handler
 *                     (CustomMsg) with feature extraction disabled, src/preprocess.cpp:337-713: the handler's filters (they are not
 *                     the cutting functions' in every detail: velodyne_handler has no ring test), no time sort, no cut - ONE frame
 *                     holding pl_surf in input order, its first point included, begin_time_s = header.stamp; hand such a frame to
 *                     lii_scan_register with scan_sorted = 0.  An empty cloud yields no frame (the node skips it, laserMapping.cpp:909-914).
 * `data` is sensor_msgs/PointCloud2::data (or the CustomPoint array) as received; the field offsets are what
 * pcl::fromROSMsg derives from msg->fields for the point structs of src/preprocess.h:35-116 (types per lidar_type:
 * VELO time f32 [s] / ring u16; OUSTER t u32 [ns] / ring u8; PANDAR timestamp f64 / ring u16; ROBOSENSE timestamp f64 /
 * ring u16 / intensity u8; intensity is not carried — nothing on the path reads it).  Decode, blind / NaN / ring /
 * point_filter_num filters, azimuth time synthesis for clouds without per-point time (:163-185), time sort and the
 * sub-frame cut (:296-334) run on the device; the frames stay there.  frames[k].begin_time_s is the value the caller
 * pushes to time_buffer (time_lidar / 1000), offset/count index the handle's frame buffer.
 * Equal time stamps keep their input order (the reference's std::sort leaves that order unspecified).
 * lii_frame_select: frame k becomes the current scan (what lidar_buffer.front() -> meas.lidar -> lii_scan_upload did). */
enum { LII_LIDAR_AVIA = 1, LII_LIDAR_VELO = 2, LII_LIDAR_OUSTER = 3, LII_LIDAR_L515 = 4, LII_LIDAR_PANDAR = 5,
       LII_LIDAR_ROBOSENSE = 6 }; /* LID_TYPE, include/common_lib.h:55 */
typedef struct lii_pc2_fields {
  int32_t point_step; /* bytes per point record */
  int32_t x, y, z, intensity, time, ring; /* byte offsets */
} lii_pc2_fields;
typedef struct lii_livox_fields {
  int32_t point_step;
  int32_t offset_time, x, y, z, reflectivity, tag, line; /* byte offsets (livox_ros_driver/CustomPoint.msg) */
} lii_livox_fields;
typedef struct lii_ingest_opts {
  uint32_t struct_size;     /* sizeof(lii_ingest_opts) */
  int32_t lidar_type;       /* LII_LIDAR_* (preprocess/lidar_type) */
  int32_t n_scans;          /* N_SCANS (preprocess/scan_line) */
  int32_t point_filter_num; /* point_filter_num */
  double blind;             /* preprocess/blind [m] */
  double stamp_s;           /* msg->header.stamp.toSec() */
  int32_t cut_frame_num;    /* required_frame_num (initialization/cut_frame_num), <= 64; 0: do not cut (initialization/cut_frame: false) */
  int32_t scan_count;       /* the caller's scan_count: the first 20 (PointCloud2) / 5 (Livox) messages are not cut */
} lii_ingest_opts;
typedef struct lii_frame_info {
  double begin_time_s;   /* time_lidar / 1000 */
  double last_offset_ms; /* curvature of the frame's last point: lidar_end_time = begin + this / 1000 (laserMapping.cpp:452-456) */
  int32_t offset, count;
} lii_frame_info;
int lii_ingest_pcl2(lii_handle h, const void* data, int32_t n_points, const lii_pc2_fields* fields, const lii_ingest_opts* opts,
                    lii_frame_info* frames, int32_t max_frames, int32_t* n_frames);
int lii_ingest_livox(lii_handle h, const void* points, int32_t n_points, const lii_livox_fields* fields,
                     const lii_ingest_opts* opts, lii_frame_info* frames, int32_t max_frames, int32_t* n_frames);
int lii_frame_select(lii_handle h, int32_t frame);
/* The overlapped forms (ABI 9) - the reference's message queue (lidar_buffer / time_buffer filled by the callbacks, src/laserMapping.cpp:326-379,
 * drained by sync_packages, :432-480) kept ON THE DEVICE: lii_ingest_*_begin puts a message under way (H2D of the raw bytes on a copy stream,
 * decode ... cut on a stream of its own) and returns without waiting; lii_ingest_end waits for the OLDEST message under way (long
 * done when it overlapped a registration), makes its frames the ones lii_frame_select serves and returns its frame table exactly as the
 * one-call form does.  Up to two messages may be under way (LII_ERR_STATE for a third); the frames of the message before stay selectable until
 * lii_ingest_end.  Call order of a single-threaded host: lii_ingest_end(m) -> register the sub-frames of m -> lii_ingest_*_begin(m + 2): the
 * begin's launches are then enqueued while the device still runs the map update of m's last sub-frame.  `data` must stay valid until the matching lii_ingest_end; from page-locked memory the transfer is asynchronous, from pageable
 * memory the call returns once the runtime has staged the bytes (the launches still overlap).  Results are the one-call forms' bits. */
int lii_ingest_pcl2_begin(lii_handle h, const void* data, int32_t n_points, const lii_pc2_fields* fields, const lii_ingest_opts* opts);
int lii_ingest_livox_begin(lii_handle h, const void* points, int32_t n_points, const lii_livox_fields* fields, const lii_ingest_opts* opts);
int lii_ingest_end(lii_handle h, lii_frame_info* frames, int32_t max_frames, int32_t* n_frames);

/* ---------------------------------------------------------------- scan-to-map registration
 * lii_iekf_iterate: ONE pass of the per-point loop + Jacobian + normal-equation reduction at a fixed state —
 *   pointBodyToWorld :209-220, Nearest_Search :980, esti_plane :997, residual/selection :987-1011,
 *   Jacobian rows :1035-1071, H^T R^-1 H / H^T R^-1 z :1073-1080 (src/laserMapping.cpp).
 *   out91 = upper triangle (78) of the 12x12, then 12, then effect_feat_num.  With a communicator attached
 *   (lii_comm_init) out91 is the all-reduced sum over ranks.
 * lii_iekf_update: the whole loop :957-1134 including the 24-state solve (:1081-1087), convergence / rematch
 *   logic (:1093-1106) and covariance update (:1109-1131).  The loop is device resident: every pass is enqueued up
 *   front, the kernels consult a control block in HBM and skip the passes the reference's schedule does not run, the
 *   final state lands in mapped host memory followed by a sequence word the call waits for — one host round trip per
 *   call, and the call returns as soon as the result exists: passes still queued behind the stopping one (they read a flag
 *   and return) drain in stream order while the caller goes on; every later call on the handle is ordered behind them.
 *   (LII_TEST=sync_result waits for the whole stream instead.  LII_TEST=host_solve drives the loop from the host
 *   around lii_iekf_iterate with the literal two-inversion algebra.)
 * lii_neighbors_download: Nearest_Points of the last search (for map_incremental, :525-549);
 *   pts = n_down x 5 x 3 floats, counts = n_down. */
int lii_iekf_iterate(lii_handle h, const lii_state* state, int32_t search, int32_t imu_en, double out91[91]);
int lii_iekf_update(lii_handle h, lii_state* state, const lii_state* state_propagated, const lii_iekf_opts* opts,
                    lii_iekf_report* report);
int lii_neighbors_download(lii_handle h, float* pts, int32_t* counts, uint8_t* selected, int32_t capacity);
/* How the last device-resident update solved its passes: *pivoted_passes = the passes (of the first 16) whose 12 x 12 elimination left the
 * pivot-free form for the routine with row exchanges (its element growth exceeded 2^8, or a pivot was zero).  The reference inverts
 * with Eigen's partial-pivoting LU every time (src/laserMapping.cpp:1081-1085); the result is held to the same tolerance either way -
 * the count exists so that tests can tell which routine they exercised. */
int lii_last_solve_info(lii_handle h, int32_t* pivoted_passes);
/* How many queries the most recent search pass could not finish inside its own launch - their 5th neighbour lies beyond what the
 * 3 x 3 x 3 cells around the query can prove: the reference's tree walks on into farther boxes for them (include/ikd-Tree/
 * ikd_Tree.cpp:827-842) - and left to the fit launch behind it (ABI 8).  Up to 256 per launch are finished by completion workgroups of
 * their own; beyond that a launch of its own finishes the listed ones (up to 4096, one wavefront each) when the scan before was in that
 * regime too, and every workgroup finishes its own points' otherwise.  ABI 9: behind lii_iekf_update / lii_scan_register on one rank the value
 * is the LARGEST count among the update's search passes and comes with the result (no device read); otherwise the most recent search
 * launch's (one small device read, synchronises the handle's stream). */
int lii_last_unfinished_queries(lii_handle h, int32_t* n_last);

/* The per-scan sequence of main() (src/laserMapping.cpp:909-1134) in ONE call, enqueued back to back on the handle's
 * stream with a single host round trip at the end: p_imu->Process' undistortion (:909; the scan is the one handed over by
 * lii_scan_upload / lii_scan_set_device, or job->scan_dev), downSizeFilterSurf.filter (:917-919) and the iterated update
 * (:957-1134).  The undistortion takes its end pose and extrinsic from `state` (the propagated state, as the reference
 * does: IMU_Processing.hpp:404-407 reads state_inout).  Equivalent to lii_undistort_* + lii_downsample(_skip) +
 * lii_iekf_update; it exists because every separate call costs the caller a host round trip. */
typedef struct lii_scan_job {
  uint32_t struct_size;            /* sizeof(lii_scan_job) */
  int32_t undistort;               /* 0 none, 1 IMU back-propagation, 2 constant-velocity model (LO mode) */
  const lii_pose6d* imu_poses;     /* undistort == 1: IMUpose table */
  int32_t n_imu_poses;
  float leaf;                      /* voxel-grid leaf size; <= 0: use the scan unfiltered */
  lii_iekf_opts opts;
  const void* scan_dev;            /* optional: adopt this device-resident scan first (as lii_scan_set_device would: float4
                                      x, y, z, t_ms; caller-owned, only read) - saves the separate call and a launch */
  int32_t n_scan_dev;
  int32_t scan_sorted;             /* 1: the scan's points are in ascending time order - the order the reference's preprocess hands
                                      every scan over in (src/preprocess.cpp:296-302) and lii_ingest_* / lii_frame_select deliver.
                                      The time-earliest point is then the first and the sweep ends with the last: the library
                                      skips the reduction that finds them (one launch per scan) and de-skews scan_dev in place of
                                      copying it first.  0: nothing is assumed.  A job that claims an order the scan does not have
                                      gets the A3 quirk / the CV sweep end applied to the wrong point - nothing else depends on it. */
  int32_t map_update;              /* 1: lii_map_incremental(h, state, NULL, NULL) with the update's final state follows the update inside
                                      this call (src/laserMapping.cpp:1146 behind :1134) - its launches are enqueued behind the update's passes
                                      while the device still works on them, instead of after the result has come back.  The caller does NOT
                                      call lii_map_incremental for this scan.  0 (the value of the formerly reserved field): nothing follows. */
  /* ABI 8 (struct_size 72; jobs of size 56 and 48 are still accepted): THE NEXT SCAN, if the caller already holds it in device memory.
   * The library then enqueues the next call's first launch - IMU de-skew + the voxel filter's insert over next_scan_dev - behind this
   * call's passes, where it waits ON THE DEVICE for what only the next call can know (the propagated state and the IMU pose table:
   * they depend on this call's result; src/laserMapping.cpp:905-915 - p_imu->Process runs on the state the update left).  The next
   * lii_scan_register hands them over with a store to mapped memory instead of a launch: the round trip result -> host -> first
   * kernel of the next scan loses the launch path (~5 us of ~10 per scan on an MI355X).  Only used when the next call asks for the
   * same thing (scan_dev == next_scan_dev, n_scan_dev == next_n_scan, undistort 1, scan_sorted 1, the same leaf, up to 64 poses);
   * anything else - another scan, any other entry point of the handle - ends the waiting launch first (it has touched nothing), and
   * a launch nobody comes for ends itself after 2 s (LII_PREARM_TIMEOUT_MS).  NULL / 0: no announcement (the forms of ABI <= 7). */
  const void* next_scan_dev;
  int32_t next_n_scan;
  int32_t reserved1;
  /* ABI 9 (struct_size 88; jobs of size 72, 56 and 48 are still accepted): THE HOST'S TIME INSIDE THE CALL.  Once every launch of the job is
   * enqueued the calling thread has nothing to do but wait for the result (~ 130 us per scan): while_waiting(while_waiting_arg) is called
   * exactly then, once, before the wait - the place for what the reference does between two scans (ros::spinOnce, src/laserMapping.cpp:893:
   * the callbacks that queue the next driver message - here lii_ingest_pcl2_begin / lii_ingest_livox_begin, whose ~ 50 us of enqueueing
   * then cost the loop nothing).  The hook may call the overlapped ingest's begin functions of this handle and anything that does not take the
   * handle; it must NOT call entry points that use the handle's stream (they would wait for, or disturb, the update under way).  A hook that
   * runs longer than the device delays the result by the difference.  Called once by every call that gets as far as its update, not by a
   * call that fails before.  NULL: nothing is called. */
  void (*while_waiting)(void* arg);
  void* while_waiting_arg;
} lii_scan_job;
int lii_scan_register(lii_handle h, const lii_scan_job* job, lii_state* state, const lii_state* state_propagated,
                      lii_iekf_report* report);

/* map_incremental (src/laserMapping.cpp:516-559): decides PointToAdd / PointNoNeedDownsample from the last
 * search's neighbour lists and applies both to the map.  n_add / n_no_downsample (the sizes of the two lists) may be NULL: then
 * the call returns without waiting for anything - the update is enqueued for predicted list sizes (those of the previous call
 * + 25 %) on a stream of its own, the next scan's arrival, de-skew and voxel filter overlap it, and whatever touches the map
 * next (a search, any lii_map_* call) waits for it first; an update whose lists outgrew the prediction is repeated there with
 * the exact sizes.  The map that results is the same either way.
 * Deferred errors: such a repeat can itself fail (LII_ERR_CAPACITY when the exact lists no longer fit max_map_points or the work
 * list - the padded predicted bounds passed the checks, the exact sizes are larger); the status then comes back from the call
 * that joined the update - the next lii_scan_register / lii_iekf_update / lii_map_* / lii_synchronize - not from this one; the
 * map is unchanged by the failed update and the handle stays usable. */
int lii_map_incremental(lii_handle h, const lii_state* state, int32_t* n_add, int32_t* n_no_downsample);

/* ---------------------------------------------------------------- LI-Init batch calibration evaluators
 * CalibState record (include/LI_init/LI_init.h:31-89). */
typedef struct lii_calib_state {
  double rot_end[9];
  double ang_vel[3];
  double linear_vel[3];
  double ang_acc[3];
  double linear_acc[3];
  double timestamp;
} lii_calib_state;

/* Uploads the aligned IMU / LiDAR-odometry state sequences (IMU_state_group / Lidar_state_group). */
int lii_calib_set_buffers(lii_handle h, const lii_calib_state* imu, const lii_calib_state* lidar, int32_t n);
/* Residual + analytic Jacobian + J^T J / J^T r / cost for one stage at the given parameters:
 *  stage 1: Angular_Vel_Cost_only_Rot (LI_init.h:91-117)   params = R_LI[9]                      -> 3 dof
 *  stage 2: Angular_Vel_Cost          (LI_init.h:119-159)  params = R_LI[9], b_g[3], t_d         -> 7 dof
 *  stage 3: Linear_acc_Cost           (LI_init.h:161-205)  params = R_GL0[9], b_a[3], T_IL[3], R_LI[9] -> 9 dof
 * Tangent convention: R <- Exp(delta) R.  JtJ is dof x dof row-major, Jtr is dof, cost = 0.5 sum r^2. */
int lii_calib_eval(lii_handle h, int32_t stage, const double* params, double* JtJ, double* Jtr, double* cost);

typedef struct lii_calib_result {
  double R_LI[9];
  double T_LI[3];
  double gyro_bias[3];
  double acc_bias[3];
  double grav_L0[3];
  double time_lag_2;
  int32_t iterations[3];
  double final_cost[3];
} lii_calib_result;
/* solve_Rotation_only / solve_Rot_bias_gyro / solve_trans_biasacc_grav (LI_init.cpp:317-492): the three
 * least-squares problems, host Levenberg-Marquardt around lii_calib_eval. stage 3 expects the buffers AFTER
 * the second time compensation + acc_interpolate (caller re-uploads, as LI_Initialization does at :619-623). */
int lii_calib_solve_stage(lii_handle h, int32_t stage, lii_calib_result* inout);

/* LI_Init::LI_Initialization as a whole (include/LI_init/LI_init.cpp:586-632): the signal-conditioning chain on the host
 * (mean filter + interpolation :82-125, zero-phase Butterworth :260-315, cross-correlation :160-193, time compensation
 * :195-238, central differences :127-158, acc interpolation :240-258) around the three GPU-evaluated solves.
 * lii_li_init_interpolate = downsample_interpolate_IMU; lii_li_init_run = everything after it. */
/* LI_Init::data_sufficiency_assess (include/LI_init/LI_init.cpp:506-556; caller src/laserMapping.cpp:1169-1177) without the
 * progress bars: omg = the LiDAR angular velocity of every LO frame so far (n_frames x 3, frame order),
 * data_accum_length = initialization/data_accum_length.  eigenvalues: of sum [w]x^T [w]x, ascending; rot_percent: the
 * pairwise products of the scaled eigenvalues (the reference's Rot_percent up to its axis order); sufficient: all > 0.99. */
int lii_data_sufficiency(const double* omg, int32_t n_frames, double data_accum_length, double eigenvalues[3],
                         double rot_percent[3], int32_t* sufficient);
int lii_li_init_interpolate(const lii_calib_state* imu_all, int32_t n_imu, const lii_calib_state* lidar, int32_t n_lidar,
                            double move_start_time, lii_calib_state* imu_out, lii_calib_state* lidar_out, int32_t* n_out);
int lii_li_init_run(lii_handle h, const lii_calib_state* imu, const lii_calib_state* lidar, int32_t n, int32_t orig_odom_freq,
                    int32_t cut_frame_num, lii_calib_result* out, double* time_lag_1, double* total_time_lag);
/* The two pieces of that chain with real arithmetic volume, on the device, bit-identical to the host
 * path of lii_li_init_run:
 *   lii_zero_phase_filter  LI_Init::zero_phase_filt (include/LI_init/LI_init.cpp:260-315; Butterworth coefficients
 *                          LI_init.h:218-224) on n_seq sequences of n CalibStates laid back to back: the four 3-vectors are
 *                          filtered, rot_end / timestamp pass through (CalibState::operator= copies only the vectors);
 *   lii_xcorr_lag          LI_Init::xcorr_temporal_init (:160-193): lag_IMU_wtr_Lidar of the |ang_vel| series, one lane per lag;
 *   lii_li_init_set_device lii_li_init_run uses them (1) or the host functions (0, default: on the reference's committed run - 1 369 states -
 *                          the call takes 2.4 ms with the host chain and 7.4 ms with the device chain, a recursive filter being serial in time;
 *                          same bits either way). */
int lii_zero_phase_filter(lii_handle h, const lii_calib_state* in, int32_t n_seq, int32_t n, lii_calib_state* out);
int lii_xcorr_lag(lii_handle h, const lii_calib_state* imu, const lii_calib_state* lidar, int32_t n, int32_t* lag_imu_wrt_lidar);
int lii_li_init_set_device(lii_handle h, int32_t on_device);

/* ---------------------------------------------------------------- multi-GPU (points of one scan sharded across ranks)
 * One process per GPU.  Rank 0 creates an id, the caller ships the 128 bytes to the other ranks (e.g.
 * torch.distributed broadcast), every rank calls lii_comm_init with the SAME state / options per scan; afterwards
 * lii_iekf_iterate / lii_iekf_update / lii_scan_register sum the 91 normal-equation scalars (fp64, in rank order, so every
 * rank forms the bit-identical sum and takes the same decisions) over the ranks.
 * Partition (lii_comm_set_partition; default 1): every rank hands over the WHOLE scan and holds the whole map.
 *   1 - by index: the de-skew and the voxel filter run replicated (their output is bit-identical on every rank, so a voxel is never
 *       split between ranks) and rank r registers the contiguous block [n r / N, n (r + 1) / N) of the down-sampled cloud, which the
 *       filter emits in the order of the voxels' first points - a stretch of the sweep per rank.
 *   2 - by voxel (SURVEY.md section 8e "a voxel-key partition"): where the voxel filter's insert is fused into the de-skew
 *       (lii_scan_register with leaf > 0; the hashed filter is then used for every scan) every rank de-skews the scan but inserts,
 *       filters, searches and fits only the voxels whose key hashes to it (~ 1 / N of them, +- a percent, whatever the scene);
 *       lii_downsample's / lii_scan_download's view of the down-sampled cloud is then this rank's share.  Elsewhere (stand-alone
 *       lii_downsample, leaf 0) the split is by index.  A share that outgrows n / N + 25 % + 2048 points fails the update with
 *       LII_ERR_CAPACITY.  Needs a transport that carries the list exchange below - the peer-mapped mailbox or RCCL - (the host-memory
 *       mailbox stays with 1; lii_comm_describe tells).
 *   0 - the caller hands every rank its own points.
 *   Either way the sharded result equals the single-GPU result up to the re-association of the 91 sums.
 *   lii_map_incremental of a sharded job: every rank decides for ITS points, and the two insert lists (PointToAdd /
 *   PointNoNeedDownsample, src/laserMapping.cpp:516-559) are exchanged - pushed into every rank's gather area behind the mailbox
 *   slots, or gathered with two ncclAllGather calls on the RCCL transport - and put together in rank order, so that every replica of
 *   the map applies the identical batch.  On the host-memory mailbox a job split by index repeats the last search for the whole
 *   cloud instead (no exchange).
 * Transports:
 *   LII_COMM_MAILBOX       ranks of ONE node; the exchange runs inside the reduce+solve kernel (no extra launch, no
 *                          collective-library call).  Every rank keeps the slots it reads in fine-grained HBM, exported through
 *                          a HIP IPC handle: an exchange pushes the 91 sums into the peers' memory (remote stores over xGMI)
 *                          and polls local memory.  Works between processes on one device as well.
 *   LII_COMM_MAILBOX_HOST  the same exchange through a POSIX shared-memory segment registered with every rank's device (host
 *                          memory over PCIe): what LII_COMM_MAILBOX falls back to in AUTO mode when an IPC handle cannot be
 *                          exported or opened.
 *   LII_COMM_RCCL          ncclAllReduce on the handle's stream between a separate final-sum and solve launch (any topology).
 *   LII_COMM_AUTO          the HBM mailbox when all ranks meet on one node within the set-up time (20 s; LII_MAILBOX_TIMEOUT_S=<exchange>[,<set-up>]), else the
 *                          host-memory mailbox, else RCCL.
 * A rank that stops calling (error on one rank only) makes the others' next update fail with LII_ERR_COMM after
 * LII_MAILBOX_TIMEOUT_S (default 30 s; RCCL: its own watchdog); the communicator must then be re-created.  lii_comm_init == lii_comm_init_ex(..., LII_COMM_AUTO). */
enum { LII_COMM_AUTO = 0, LII_COMM_RCCL = 1, LII_COMM_MAILBOX = 2, LII_COMM_MAILBOX_HOST = 3 };
int lii_comm_unique_id(uint8_t id_out[128]);
int lii_comm_init(lii_handle h, int32_t n_ranks, int32_t rank, const uint8_t id[128]);
int lii_comm_init_ex(lii_handle h, int32_t n_ranks, int32_t rank, const uint8_t id[128], int32_t transport);
int lii_comm_transport(lii_handle h, int32_t* transport); /* the transport in use; LII_COMM_AUTO = none (single rank) */
/* Which transport this rank ended up with and why, as text (e.g. "mailbox in registered host memory - the HBM form was not
 * possible: device 0 cannot access its peer 0000:c1:00.0 (hipDeviceCanAccessPeer)"); LII_DIAG=1 prints it at set-up. */
int lii_comm_describe(lii_handle h, char* out, int32_t capacity);
int lii_comm_rccl_ranks(lii_handle h, int32_t* n_ranks);  /* ncclCommCount of the attached RCCL communicator; 0: none attached */
int lii_comm_set_partition(lii_handle h, int32_t library_partition);  /* 0 caller | 1 by index (default) | 2 by voxel */
int lii_comm_destroy(lii_handle h);
/* TEST ENTRY POINT - not part of the per-scan path.  It plays its ranks in the handle's own list buffers: LII_ERR_STATE while a map update
 * of the handle is pending (its lists would be overwritten; call lii_map_commit first); leaves the exchange's sequence number and scratch
 * as it found them.
 * Self-test of the list exchange of a sharded job's map update (lii_map_incremental; lii_exchange.hip) with n_ranks > 1 on ONE device:
 * the handle (not a rank of a job) plays rank 0 .. n_ranks - 1 in turn.  form 0: the gather areas of the mailbox transport; form 1:
 * the trimmed all-gather layout of the RCCL transport (the functions lists_exchange_rccl is made of, device copies standing in for the
 * two ncclAllGather calls - a communicator holds one rank per device, the layout can still be exercised).  Rank r's lists: n_add[r] /
 * n_nodown[r] points (x, y, z, w) behind each other in add_xyzw / nodown_xyzw.  Every played rank must end with the identical pair
 * of joined lists (else LII_ERR_COMM); they are returned (capacity: points per output).  Reference: the two Add_Points calls every
 * replica of the map must receive identically, src/laserMapping.cpp:556-557. */
int lii_selftest_list_exchange(lii_handle h, int32_t n_ranks, int32_t form, const float* add_xyzw, const int32_t* n_add, const float* nodown_xyzw,
                               const int32_t* n_nodown, float* out_add, int32_t* out_n_add, float* out_nodown, int32_t* out_n_nodown, int32_t capacity);

/* ---------------------------------------------------------------- parameter surface (config/<sensor>.yaml + launch/<sensor>.launch)
 * The nh.param<> block of main() (src/laserMapping.cpp:767-799) as a POD: same names (`section/key` -> field), same defaults.
 * roslaunch fills the parameter server from `<rosparam command="load" file="$(find lidar_imu_init)/config/X.yaml"/>` and
 * `<param name=".." value=".."/>` (launch/<sensor>.launch:6-12); a host without ROS calls
 *   lii_params_defaults    the third argument of every nh.param<> call
 *   lii_params_load_yaml   one config/<sensor>.yaml on top (block maps, scalars, quoted strings, flow lists, '#' comments)
 *   lii_params_load_launch a launch file: its rosparam yaml (looked up in `config_dir`, else <launch dir>/../config) first,
 *                          then its <param> tags (names a node never reads are ignored, as on the parameter server)
 *   lii_params_set         one override by name ("max_iteration", "mapping/filter_size_surf", ...)
 *   lii_params_apply       -> lii_config (map down-sample box = mapping/filter_size_map), lii_ingest_opts (lidar_type, scan_line,
 *                          point_filter_num, blind, cut_frame_num; stamp / scan_count stay per message), lii_iekf_opts
 *                          (max_iteration; imu_en = 0 as at start-up, laserMapping.cpp:87), voxel leaf = mapping/filter_size_surf.
 * Host functions (no device work, no handle). */
typedef struct lii_params {
  uint32_t struct_size; /* sizeof(lii_params); set by lii_params_defaults */
  int32_t max_iteration, point_filter_num;
  double filter_size_surf, filter_size_map, cube_side_length, det_range;
  double gyr_cov, acc_cov, grav_cov, b_gyr_cov, b_acc_cov;
  double blind;
  int32_t lidar_type, scan_line, feature_extract_en;
  int32_t cut_frame, cut_frame_num, orig_odom_freq;
  double online_refine_time, mean_acc_norm, data_accum_length;
  double Rot_LI_cov[3], Trans_LI_cov[3];
  int32_t n_Rot_LI_cov, n_Trans_LI_cov; /* entries present in the file (the reference reads vector<double>) */
  int32_t path_en, scan_publish_en, dense_publish_en, scan_bodyframe_pub_en, runtime_pos_log_enable, pcd_save_en, pcd_save_interval;
  int32_t reserved0;
  char lid_topic[128], imu_topic[128], map_file_path[256];
} lii_params;
int lii_params_defaults(lii_params* p);
int lii_params_load_yaml(const char* yaml_path, lii_params* inout);
int lii_params_load_launch(const char* launch_path, const char* config_dir /* may be NULL */, lii_params* inout);
int lii_params_set(lii_params* inout, const char* name, const char* value);
int lii_params_apply(const lii_params* p, int32_t device, int32_t max_scan_points, int32_t max_map_points, lii_config* cfg /* may be NULL */,
                     lii_ingest_opts* ingest /* may be NULL */, lii_iekf_opts* opts /* may be NULL */, float* leaf /* may be NULL */);
const char* lii_params_last_error(void);

/* ---------------------------------------------------------------- utilities for harnesses */
int lii_dev_alloc(lii_handle h, size_t bytes, void** dev_ptr);
int lii_dev_free(lii_handle h, void* dev_ptr);
int lii_dev_upload(lii_handle h, void* dev_dst, const void* host_src, size_t bytes);
/* Accumulated device timings [ms] since lii_set_profiling(h, 1), measured with HIP events on the handle's stream:
 * lii_iekf_update / lii_scan_register: [5] number of executed k-NN passes, [7] their total time (events around the k-NN launches
 * only: every event is a barrier on the stream), [4] wall time of the last update; lii_iekf_iterate: also [0] / [1] search-pass /
 * residual-pass kernels, [2] the final sum, [6] count of [1]; [3] host solve (host-driven loop only). */
int lii_set_profiling(lii_handle h, int32_t enabled); /* 1: start (zero the accumulators), 2: resume, 0: pause, 3: see below */
int lii_last_timings(lii_handle h, double out_ms[8]);
/* lii_set_profiling(h, 3): lii_scan_register brackets EVERY launch of the scan with HIP events (a barrier packet each: the
 * scan runs slower than unprofiled - this mode is for attributing time, not for measuring throughput) and accumulates, per
 * kind of launch, the time from its event to the next one (kernel + dispatch) and the number of launches that executed.
 * Accumulators start at lii_set_profiling(h, 1), like the others. */
enum lii_kernel_kind {
  LII_KP_DESKEW = 0,     /* adoption + de-skew (+ the voxel filter's insert, + the time-extent launch of an unsorted scan) */
  LII_KP_VOXEL = 1,      /* the rest of the voxel filter */
  LII_KP_KNN = 2,        /* k-NN pass */
  LII_KP_FIT_SEARCH = 3, /* plane fit + residual + reduction behind a k-NN pass */
  LII_KP_FIT = 4,        /* residual + reduction on cached planes */
  LII_KP_SOLVE = 5,      /* final sum + 24-state solve */
  LII_KP_KINDS = 8
};
typedef struct lii_kernel_profile {
  uint32_t struct_size; /* sizeof(lii_kernel_profile) */
  int32_t scans;        /* scans profiled */
  double ms[LII_KP_KINDS];
  int32_t launches[LII_KP_KINDS];
} lii_kernel_profile;
int lii_last_kernel_profile(lii_handle h, lii_kernel_profile* out);

#ifdef __cplusplus
}
#endif
#endif /* LIINIT_HIP_H */
