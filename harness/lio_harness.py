"""ROS-free harness for the LiDAR-inertial (LIO) phase that follows LI_Initialization (src/laserMapping.cpp:1203-1238: the
node switches `imu_en` on, re-expresses the pose in the IMU frame and keeps refining the extrinsic online).

Host side, restated in numpy for tests only (the reference keeps it on the host too, SURVEY.md §8 boundary):
  ImuProcess::propagation_and_undist, forward part (src/IMU_Processing.hpp:269-382): mid-point IMU integration between two
  scan ends, the 24-state covariance propagation F_x P F_x^T + Q, and the IMUpose table (msg/Pose6D.msg) that the
  back-propagation needs.
Device side, through the C-ABI: back-propagation de-skew (lii_undistort_imu), voxel grid, iterated update with the extrinsic
in the state (12-column H), map_incremental — via `Registrar.scan_register` + `map_incremental`.
"""
from __future__ import annotations

import numpy as np

from lidar_imu_init_amd.api import Registrar, State, pose6d_array
from .lo_harness import so3_exp

G_m_s2 = 9.81


def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


class LioOdometry:
    def __init__(self, reg: Registrar, state: State, filter_size_surf=0.1, max_iteration=5, cov_gyr=0.1, cov_acc=0.1,
                 cov_bias_gyr=1e-4, cov_bias_acc=1e-4, cov_R_LI=1e-5, cov_T_LI=1e-5, imu_mean_acc_norm=G_m_s2):
        self.reg, self.state = reg, state
        self.leaf, self.max_it = filter_size_surf, max_iteration
        self.cov_gyr, self.cov_acc = np.full(3, cov_gyr), np.full(3, cov_acc)
        self.cov_bg, self.cov_ba = np.full(3, cov_bias_gyr), np.full(3, cov_bias_acc)
        self.cov_R_LI, self.cov_T_LI = np.full(3, cov_R_LI), np.full(3, cov_T_LI)
        self.mean_acc_norm = imu_mean_acc_norm
        self.last_imu = None             # (t, gyro, accel)
        self.last_lidar_end_time = None
        self.angvel_last, self.acc_s_last = np.zeros(3), np.zeros(3)
        self.first = True
        self.reports = []

    def propagate(self, imu, pcl_beg_time, pcl_end_time):
        """imu: list of (t, gyro, accel) of this scan (stamps <= scan end).  Returns the IMUpose table (K, 22)."""
        st = self.state
        v_imu = [self.last_imu] + list(imu)
        imu_end_time = v_imu[-1][0]
        poses = [np.r_[0.0, self.acc_s_last, self.angvel_last, st.vel_end, st.pos_end, st.rot_end.reshape(-1)]]
        vel, pos, R = st.vel_end.copy(), st.pos_end.copy(), st.rot_end.copy()
        angvel_avr, acc_imu = self.angvel_last.copy(), self.acc_s_last.copy()
        for head, tail in zip(v_imu[:-1], v_imu[1:]):
            if tail[0] < self.last_lidar_end_time:
                continue
            angvel_avr = 0.5 * (head[1] + tail[1]) - st.bias_g
            acc_avr = 0.5 * (head[2] + tail[2]) / self.mean_acc_norm * G_m_s2 - st.bias_a
            dt = tail[0] - (self.last_lidar_end_time if head[0] < self.last_lidar_end_time else head[0])
            Exp_f = so3_exp(angvel_avr * dt)
            F = np.eye(24)
            F[0:3, 0:3] = so3_exp(-angvel_avr * dt)
            F[0:3, 15:18] = -np.eye(3) * dt
            F[3:6, 12:15] = np.eye(3) * dt
            F[12:15, 0:3] = -R @ skew(acc_avr) * dt
            F[12:15, 18:21] = -R * dt
            F[12:15, 21:24] = np.eye(3) * dt
            Q = np.zeros((24, 24))
            Q[0:3, 0:3] = np.diag(self.cov_gyr * dt * dt)
            Q[6:9, 6:9] = np.diag(self.cov_R_LI * dt * dt)
            Q[9:12, 9:12] = np.diag(self.cov_T_LI * dt * dt)
            Q[12:15, 12:15] = R @ np.diag(self.cov_acc) @ R.T * dt * dt
            Q[15:18, 15:18] = np.diag(self.cov_bg * dt * dt)
            Q[18:21, 18:21] = np.diag(self.cov_ba * dt * dt)
            st.cov[:] = F @ st.cov @ F.T + Q
            R = R @ Exp_f
            acc_imu = R @ acc_avr + st.gravity
            pos = pos + vel * dt + 0.5 * acc_imu * dt * dt
            vel = vel + acc_imu * dt
            self.angvel_last, self.acc_s_last = angvel_avr, acc_imu
            poses.append(np.r_[tail[0] - pcl_beg_time, acc_imu, angvel_avr, vel, pos, R.reshape(-1)])
        note = 1.0 if pcl_end_time > imu_end_time else -1.0
        dt = note * (pcl_end_time - imu_end_time)
        st.vel_end[:] = vel + note * acc_imu * dt
        st.rot_end[:] = R @ so3_exp(note * angvel_avr * dt)
        st.pos_end[:] = pos + note * vel * dt + note * 0.5 * acc_imu * dt * dt
        self.last_imu = v_imu[-1]
        self.last_lidar_end_time = pcl_end_time
        T = pose6d_array(len(poses))
        T[:] = np.array(poses)
        return T

    def process(self, scan4: np.ndarray, t_beg: float, imu):
        """scan4: raw (skewed) LiDAR-frame points with time offsets [ms]; imu: [(t, gyro, accel)] up to the scan end."""
        t_end = t_beg + float(scan4[:, 3].max()) / 1000.0
        st = self.state
        if self.last_imu is None:
            self.last_imu = imu[0]
            self.last_lidar_end_time = t_beg
        table = self.propagate(imu, t_beg, t_end)
        self.reg.scan_upload(scan4)
        if self.first:
            self.reg.undistort_imu(table, st.rot_end, st.pos_end, st.offset_R_L_I, st.offset_T_L_I)
            self.reg.downsample(self.leaf, want_count=False)
            body = self.reg.scan_download(1)[:, :3].astype(np.float64)
            p_I = body @ st.offset_R_L_I.T + st.offset_T_L_I
            self.reg.map_build((p_I @ st.rot_end.T + st.pos_end).astype(np.float32))
            self.first = False
            return None
        prop = st.copy()
        rep = self.reg.scan_register(st, prop, imu_poses=table, leaf=self.leaf, max_iterations=self.max_it, imu_en=True)
        self.reg.map_incremental(st)
        self.reports.append(rep)
        return rep
