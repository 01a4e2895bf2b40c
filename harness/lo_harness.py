"""ROS-free `laserMapping`-shaped harness for the LiDAR-only-odometry phase (LO mode) and the hand-over to LI_init.

This is the plumbing around the hot path that the reference keeps on the host (src/laserMapping.cpp:891-1238): per scan
  ImuProcess::Forward_propagation_without_imu (constant-velocity propagation of pose + covariance,
      src/IMU_Processing.hpp:204-244)                                  -> `cv_propagate` (numpy, host)
  CV de-skew, voxel filter, iterated update, map_incremental           -> libliinit_hip through `Registrar`
  Init_LI->push_Lidar_CalibState(rot_end, bias_g, vel_end, t)          -> `LoOdometry.lidar_states`
It exists for end-to-end tests on synthetic streams (BASELINE.json configs[0]-style plumbing); nothing here is timed.
"""
from __future__ import annotations

import numpy as np

from lidar_imu_init_amd.api import Registrar, State, calib_state_array


def so3_exp(w):
    th = np.linalg.norm(w)
    if th < 1e-12:
        return np.eye(3)
    k = w / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def cv_propagate(st: State, dt: float, cov_gyr: float, cov_acc: float):
    """Forward_propagation_without_imu, :224-244: in LO mode `bias_g` holds the angular velocity and `vel_end` the linear
    velocity (CV model).  F_x(0,0) = Exp(w, -dt), F_x(0,15) = I dt, F_x(3,12) = I dt; Q(15,15) = cov_gyr dt^2, Q(12,12) = cov_acc dt^2."""
    w = st.bias_g.copy()
    F = np.eye(24)
    F[0:3, 0:3] = so3_exp(-w * dt)
    F[0:3, 15:18] = np.eye(3) * dt
    F[3:6, 12:15] = np.eye(3) * dt
    Q = np.zeros((24, 24))
    Q[15:18, 15:18] = np.eye(3) * cov_gyr * dt * dt
    Q[12:15, 12:15] = np.eye(3) * cov_acc * dt * dt
    st.cov[:] = F @ st.cov @ F.T + Q
    st.rot_end[:] = st.rot_end @ so3_exp(w * dt)
    st.pos_end[:] = st.pos_end + st.vel_end * dt


class LoOdometry:
    """FAST-LO: scan-to-map registration with a constant-velocity motion model on the GPU path."""

    def __init__(self, reg: Registrar, filter_size_surf=0.1, max_iteration=5, gyr_cov=50.0, acc_cov=2.0):
        self.reg = reg
        self.leaf = filter_size_surf
        self.max_it = max_iteration
        self.gyr_cov, self.acc_cov = gyr_cov, acc_cov
        self.state = State()
        self.first = True
        self.t_last_beg = None
        self.lidar_states = []  # (rot_end, ang_vel, linear_vel, lidar_end_time)
        self.positions = []     # pos_end after every update (diagnostics)
        self.reports = []

    def process(self, scan4: np.ndarray, t_beg: float):
        """scan4: (n,4) float32 body-frame points with per-point time offsets [ms] from t_beg."""
        self.reg.scan_upload(scan4)
        return self.process_current(t_beg, t_beg + float(scan4[:, 3].max()) / 1000.0)

    def process_current(self, t_beg: float, t_end: float):
        """The scan is already the handle's current scan (lii_scan_upload or lii_frame_select after a device ingest)."""
        st = self.state
        dt = 0.1 if self.t_last_beg is None else (t_beg - self.t_last_beg)  # b_first_frame_ -> 0.1 (:215-221)
        self.t_last_beg = t_beg
        cv_propagate(st, dt, self.gyr_cov, self.acc_cov)
        self.reg.undistort_cv(st.bias_g, st.vel_end, st.rot_end)
        self.reg.downsample(self.leaf, want_count=False)
        if self.first:
            body = self.reg.scan_download(1)
            world = body[:, :3].astype(np.float64) @ st.rot_end.T + st.pos_end
            self.reg.map_build(world.astype(np.float32))  # ikdtree.Build on the first scan (:921-931)
            self.first = False
            return None
        prop = st.copy()
        rep = self.reg.iekf_update(st, prop, max_iterations=self.max_it, imu_en=False)
        self.reg.map_incremental(st)
        self.reports.append(rep)
        self.lidar_states.append((st.rot_end.copy(), st.bias_g.copy(), st.vel_end.copy(), t_end))
        self.positions.append(st.pos_end.copy())
        return rep

    def lidar_calib_states(self):
        a = calib_state_array(len(self.lidar_states))
        for i, (R, w, v, t) in enumerate(self.lidar_states):
            a[i, 0:9] = R.reshape(-1)
            a[i, 9:12] = w
            a[i, 12:15] = v
            a[i, 21] = t
        return a
