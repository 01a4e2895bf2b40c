"""The de-skew restatement (oracle/orc_scan.hpp) and the numpy IMU forward propagation of the test harness
(harness/lio_harness.py) held to the UNMODIFIED reference src/IMU_Processing.hpp, compiled where it lies under /root/reference
against oracle/ref_shim_imu + oracle/ref_shim_math into oracle/_ref/libref_imu.so (oracle/ref_imu_wrap.cpp drives
ImuProcess::Process in both modes).

  * LIO mode (propagation_and_undist, :271-417): the reference propagates the state over the scan's IMU samples, leaves the
    IMUpose table behind and back-propagates every point.  The oracle's undistort_imu, given the reference's own pose table and
    propagated pose, must reproduce the de-skewed cloud BIT FOR BIT - including quirk A3 (the time-earliest point is compensated
    once per pose it lies behind, reading its float32 coordinates back in between) and the points at or before the first pose,
    which stay untouched.
  * LO mode (Forward_propagation_without_imu, :207-269): constant-velocity propagation, then every point but the time-earliest
    one is rotated / shifted to the scan end (quirk A2 / A3) - bit for bit as well.
  * the harness's numpy forward propagation against the reference's: pose table and state mean to 1e-12 (numpy's rotation
    composition rounds differently in the last bit), covariance to 1e-12 of its largest entry (the shim multiplies the 24 x 24
    matrices as plain triple loops; Eigen's own blocked product may sum in yet another order).
CPU only.  Skipped where oracle/_ref was never built (the GPU box: /root/reference does not exist there).
"""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.skipif(O.ref_imu_lib() is None, reason="oracle/_ref/libref_imu.so not built (no /root/reference)")


def _rot(rx, ry, rz):
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def _state(rng, moving=True):
    st = O.state_init()
    v = O.StateView(st)
    v.rot_end[:] = _rot(0.03, -0.02, 0.7)
    v.pos_end[:] = [1.5, -0.7, 0.3]
    v.offset_R_L_I[:] = _rot(0.01, 0.02, -0.015)
    v.offset_T_L_I[:] = [0.04, 0.02, -0.03]
    v.vel_end[:] = [1.2, -0.4, 0.1] if moving else 0
    v.bias_g[:] = [0.002, -0.001, 0.0015]
    v.bias_a[:] = [0.01, 0.02, -0.015]
    v.gravity[:] = [0.05, -0.03, -9.8]
    A = rng.normal(size=(24, 24)) * 1e-2
    v.cov[:] = A @ A.T + np.eye(24) * 1e-3
    return st


def _imu_samples(rng, t0, t1, rate=200.0):
    ts = np.arange(t0, t1 + 0.5 / rate, 1.0 / rate)
    rows = []
    for t in ts:
        gyr = np.array([0.4 * np.sin(7 * t), -0.3 * np.cos(5 * t), 0.8 + 0.2 * np.sin(3 * t)]) + rng.normal(size=3) * 1e-3
        acc = np.array([0.5 * np.cos(4 * t), 0.3 * np.sin(6 * t), 9.7 + 0.2 * np.cos(2 * t)]) + rng.normal(size=3) * 1e-2
        rows.append(np.r_[t, gyr, acc])
    return np.array(rows)


def _cloud(rng, n, t_lo_ms, t_hi_ms, zero_time_points=0):
    pts = np.zeros((n, 4), np.float32)
    pts[:, :3] = rng.uniform(-40, 40, size=(n, 3))
    t = np.sort(rng.choice(np.arange(int(t_lo_ms * 1000), int(t_hi_ms * 1000)), size=n, replace=False)) / 1000.0  # distinct stamps
    pts[:, 3] = t.astype(np.float32)
    assert len(np.unique(pts[:, 3])) == n
    if zero_time_points:
        pts[:zero_time_points, 3] = 0.0  # at the first pose: untouched by the reference (strict '>')
    return pts[rng.permutation(n)]  # the reference sorts by time itself


@pytest.mark.parametrize("t_first_ms", [0.004, 31.7])  # 31.7 ms: the earliest point lies behind six poses (quirk A3)
def test_lio_deskew_bit_for_bit_and_forward_propagation(t_first_ms):
    rng = np.random.default_rng(11)
    st = _state(rng)
    beg = 100.0
    imu = _imu_samples(rng, beg + 0.002, beg + 0.0995)
    last_imu = np.r_[beg - 0.003, 0.1, -0.2, 0.7, 0.2, 0.1, 9.75]
    pts = _cloud(rng, 6000, t_first_ms, 100.0, zero_time_points=1 if t_first_ms < 1 else 0)
    acc_s_last, angvel_last = np.array([0.1, -0.05, 0.02]), np.array([0.3, -0.1, 0.75])
    ref = O.ref_imu_process_lio(imu, last_imu, beg - 0.0004, acc_s_last, angvel_last, np.full(3, 0.1), np.full(3, 0.1), 9.79, beg, st, pts)
    poses, sp = ref["poses"], O.StateView(ref["state"])
    assert len(poses) == len(imu) + 1 and poses[0, 0] == 0.0
    if t_first_ms > 1:
        assert (poses[:, 0] < t_first_ms / 1000.0).sum() >= 6
    # --- the oracle's de-skew on the reference's own table and propagated pose
    got = O.undistort_imu(pts, poses, sp.rot_end, sp.pos_end, sp.offset_R_L_I, sp.offset_T_L_I)
    assert np.array_equal(got[:, 3], ref["points"][:, 3])          # same (time-sorted) order
    assert np.array_equal(got.view(np.uint32), ref["points"].view(np.uint32)), "de-skew differs from the reference's"
    moved = np.any(ref["points"][:, :3] != pts[np.argsort(pts[:, 3], kind="stable")][:, :3], axis=1)
    assert moved.sum() >= len(pts) - 1 and (t_first_ms > 1 or not moved[0])  # (the point at t = 0 stays where it was: strict ">")
    # --- the harness's numpy forward propagation against the reference's
    from harness.lio_harness import LioOdometry
    import lidar_imu_init_amd as lii
    s = lii.State(st)
    odo = LioOdometry(None, s, cov_gyr=0.1, cov_acc=0.1, cov_R_LI=1e-5, cov_T_LI=1e-4, imu_mean_acc_norm=9.79)  # (the constructor defaults of ImuProcess, :96-101)
    odo.last_imu = (last_imu[0], last_imu[1:4].copy(), last_imu[4:7].copy())
    odo.last_lidar_end_time = beg - 0.0004
    odo.acc_s_last, odo.angvel_last = acc_s_last.copy(), angvel_last.copy()
    t_end = beg + float(np.float32(pts[:, 3].max())) / 1000.0
    table = odo.propagate([(r[0], r[1:4].copy(), r[4:7].copy()) for r in imu], beg, t_end)
    assert table.shape == poses.shape
    assert np.max(np.abs(np.asarray(table) - poses)) <= 1e-12 * max(1.0, np.max(np.abs(poses)))
    assert np.max(np.abs(s.rot_end - sp.rot_end)) <= 1e-13 and np.max(np.abs(s.pos_end - sp.pos_end)) <= 1e-12
    assert np.max(np.abs(s.vel_end - sp.vel_end)) <= 1e-12
    assert np.max(np.abs(s.cov - sp.cov)) <= 1e-12 * np.max(np.abs(sp.cov))
    assert abs(odo.last_lidar_end_time - ref["last_lidar_end_time"]) == 0.0
    assert np.max(np.abs(odo.acc_s_last - ref["acc_s_last"])) <= 1e-12 and np.max(np.abs(odo.angvel_last - ref["angvel_last"])) <= 1e-15


@pytest.mark.parametrize("first_frame", [True, False])
def test_cv_deskew_bit_for_bit(first_frame):
    rng = np.random.default_rng(5)
    st = _state(rng)
    pts = _cloud(rng, 5000, 0.003, 99.9)
    beg, last = 50.0, 49.9
    st_ref, pts_ref = O.ref_imu_process_cv(beg, last, first_frame, np.full(3, 0.1), np.full(3, 0.2), st, pts)
    v0, v1 = O.StateView(st), O.StateView(st_ref)
    dt = 0.1 if first_frame else beg - last
    # constant-velocity propagation of the mean (:236-240): rot_end <- rot_end Exp(bias_g dt), pos_end += vel_end dt
    assert np.max(np.abs(v1.rot_end - v0.rot_end @ O.exp_so3(v0.bias_g, dt))) <= 1e-15
    assert np.max(np.abs(v1.pos_end - (v0.pos_end + v0.vel_end * dt))) <= 1e-15
    got = O.undistort_cv(pts, v1.bias_g, v1.vel_end, v1.rot_end)
    assert np.array_equal(got.view(np.uint32), pts_ref.view(np.uint32)), "CV de-skew differs from the reference's"
    srt = pts[np.argsort(pts[:, 3], kind="stable")]
    # quirk A3: the earliest point stays where it was (the loop stops ahead of it); so does the latest one (dt_j = 0)
    assert np.array_equal(pts_ref[0], srt[0]) and np.all(np.any(pts_ref[1:-1, :3] != srt[1:-1, :3], axis=1))
