"""The oracle's per-scan point operations (oracle/orc_scan.hpp) against INDEPENDENTLY written numpy / scipy versions of the same
published algorithms - the checks of these rows (a1, a2, a5) that do not go through the oracle's own code:
  * pcl::VoxelGrid<PointXYZINormal>::filter (PCL 1.8 voxel_grid.hpp as documented: getMinMax3D over the finite points, the
    int32 index-overflow guard -> identity, min_b / div_b, linear index, sort, centroid of ALL fields, ascending-index output;
    call site src/laserMapping.cpp:917-919) - bit for bit (same float32 sums in input order);
  * the IMU back-propagation loop of ImuProcess::propagation_and_undist (src/IMU_Processing.hpp:390-414), written per POINT with
    a search for its head pose instead of the reference's backwards walk, scipy rotations for Exp - to float32 rounding;
    including the time-earliest point being compensated once per qualifying head and points at or before
    the first pose staying untouched;
  * the CV de-skew of Forward_propagation_without_imu (src/IMU_Processing.hpp:246-266), vectorised.
PCL and the reference's ROS translation units cannot be built here; the device kernels are tested bit-exactly against the
oracle (tests/test_gpu_scan_ops.py), so these checks extend to them."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation


def _voxel_grid_numpy(pts, leaf):
    """Returns (out, filtered) like oracle.voxel_grid."""
    pts = np.ascontiguousarray(pts, np.float32)
    if len(pts) == 0:
        return pts.copy(), True
    inv = np.float32(1.0) / np.float32(leaf)
    finite = np.isfinite(pts[:, :3]).all(axis=1)
    xyz = pts[finite, :3]
    mn, mx = xyz.min(axis=0), xyz.max(axis=0)
    d = ((mx - mn) * inv).astype(np.int64) + 1          # truncation toward zero of a non-negative float, as the int64 cast does
    if int(d[0]) * int(d[1]) * int(d[2]) > np.iinfo(np.int32).max:
        return pts.copy(), False
    min_b = np.floor(mn * inv).astype(np.int32)
    max_b = np.floor(mx * inv).astype(np.int32)
    div_b = max_b - min_b + 1
    mul = np.array([1, div_b[0], div_b[0] * div_b[1]], np.int32)
    ijk = (np.floor(xyz * inv) - min_b.astype(np.float32)).astype(np.int32)
    lin = (ijk * mul).sum(axis=1).astype(np.int32)
    src = np.nonzero(finite)[0]
    order = np.lexsort((src, lin))                       # by voxel, then by input position
    lin_s, src_s = lin[order], src[order]
    starts = np.r_[0, np.nonzero(np.diff(lin_s))[0] + 1, len(lin_s)]
    out = np.zeros((len(starts) - 1, 4), np.float32)
    for g in range(len(starts) - 1):
        members = pts[src_s[starts[g]:starts[g + 1]]]
        s = np.cumsum(members, axis=0, dtype=np.float32)[-1]   # sequential float32 sums in input order
        out[g] = s / np.float32(len(members))
    return out, True


@pytest.mark.parametrize("n,leaf,spread,seed", [(6000, 0.5, 8.0, 1), (20000, 0.05, 3.0, 2), (3000, 0.2, 40.0, 3), (1, 0.1, 1.0, 4)])
def test_voxel_grid_against_independent_numpy(oracle, n, leaf, spread, seed):
    rng = np.random.default_rng(seed)
    pts = np.c_[rng.normal(0, spread, (n, 3)), rng.uniform(0, 100, n)].astype(np.float32)
    if n > 100:
        pts[5, 0] = np.nan           # non-finite points are skipped (the cloud is not dense)
        pts[9, 2] = np.inf
        pts[20:30] = pts[10:20]      # exact duplicates share a voxel
        pts[40:60, :3] = np.floor(pts[40:60, :3] / leaf) * np.float32(leaf)   # points exactly on voxel faces
    got, f_got = oracle.voxel_grid(pts, leaf)
    ref, f_ref = _voxel_grid_numpy(pts, leaf)
    assert f_got == f_ref and got.shape == ref.shape
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_voxel_grid_overflow_identity_against_independent_numpy(oracle):
    """dx * dy * dz beyond int32: PCL warns and copies the input (voxel_grid.hpp) - the identity path lii_downsample_skip names."""
    rng = np.random.default_rng(7)
    pts = np.c_[rng.uniform(-500, 500, (2000, 3)), rng.uniform(0, 100, 2000)].astype(np.float32)
    got, f_got = oracle.voxel_grid(pts, 0.01)
    ref, f_ref = _voxel_grid_numpy(pts, 0.01)
    assert not f_got and not f_ref
    assert np.array_equal(got, ref) and np.array_equal(got, pts)


def _exp(w, dt):
    return Rotation.from_rotvec(np.asarray(w, float) * dt).as_matrix()


def _undistort_imu_numpy(pts_sorted, poses, end_R, end_p, R_LI, T_LI):
    """pts_sorted: time-sorted float32 (n,4); poses: (K,22).  Per-point formulation of IMU_Processing.hpp:390-414."""
    out = pts_sorted.astype(np.float32).copy()
    K = len(poses)
    if len(out) == 0 or K < 2:
        return out
    t = out[:, 3].astype(np.float64) / 1000.0
    off = poses[:, 0]

    def compensate(p_in, h, tj):
        acc, gyr, vel, pos, rot = poses[h, 1:4], poses[h, 4:7], poses[h, 7:10], poses[h, 10:13], poses[h, 13:22].reshape(3, 3)
        dt = tj - off[h]
        R_i = rot @ _exp(gyr, dt)
        P_i = pos + vel * dt + 0.5 * acc * dt * dt
        return R_LI.T @ (end_R.T @ (R_i @ (R_LI @ p_in + T_LI) + P_i - end_p) - T_LI)

    heads = np.arange(K - 1)                               # candidate heads 0 .. K-2
    for j in range(1, len(out)):
        q = heads[off[:K - 1] < t[j]]                      # strict: a point AT a pose time belongs to the earlier head
        if len(q) == 0:
            continue                                       # at or before the first pose: the loop never reaches it
        out[j, :3] = compensate(out[j, :3].astype(np.float64), q[-1], t[j]).astype(np.float32)
    # the time-earliest point ends the inner loop of EVERY head that reaches it: it is compensated once per head whose offset
    # time lies below its own time, latest head first, each time from its already rewritten (float32) coordinates
    for h in heads[off[:K - 1] < t[0]][::-1]:
        out[0, :3] = compensate(out[0, :3].astype(np.float64), h, t[0]).astype(np.float32)
    return out


@pytest.mark.parametrize("n,K,t0,seed", [(4000, 12, 0.0, 1), (800, 3, 0.6, 2), (1, 5, 2.0, 3), (300, 2, 0.0, 4)])
def test_undistort_imu_against_independent_numpy(oracle, n, K, t0, seed):
    rng = np.random.default_rng(seed)
    poses = np.zeros((K, 22))
    poses[:, 0] = np.sort(np.r_[0.0, rng.uniform(0.003, 0.1, K - 1)])
    R = np.eye(3)
    for k in range(K):
        poses[k, 1:4] = rng.normal(0, 1.0, 3)
        poses[k, 4:7] = rng.normal(0, 0.8, 3)
        poses[k, 7:10] = rng.normal(0, 0.5, 3)
        poses[k, 10:13] = rng.normal(0, 0.05, 3)
        R = R @ Rotation.from_rotvec(rng.normal(0, 0.02, 3)).as_matrix()
        poses[k, 13:22] = R.reshape(-1)
    pts = np.c_[rng.uniform(-30, 30, (n, 3)), rng.uniform(t0, 100.0, n)].astype(np.float32)
    if n > 10:
        pts[3, 3] = np.float32(poses[1, 0] * 1000.0)       # a point exactly at a pose time
        pts[4, 3] = 0.0                                     # and one at the scan start
    end_R = Rotation.from_rotvec(rng.normal(0, 0.05, 3)).as_matrix()
    end_p = rng.normal(0, 0.2, 3)
    R_LI = Rotation.from_rotvec(rng.normal(0, 0.3, 3)).as_matrix()
    T_LI = rng.normal(0, 0.1, 3)
    got = oracle.undistort_imu(pts, poses, end_R, end_p, R_LI, T_LI)
    srt = pts[np.argsort(pts[:, 3], kind="stable")]
    assert np.array_equal(got[:, 3], srt[:, 3])
    ref = _undistort_imu_numpy(srt, poses, end_R, end_p, R_LI, T_LI)
    assert np.allclose(got[:, :3], ref[:, :3], rtol=0, atol=2e-5)   # |p| <= 60 m in float32: 1 - 2 ulp
    # what must NOT move: everything at or before the first pose's time
    still = srt[:, 3].astype(np.float64) / 1000.0 <= poses[0, 0]
    still[0] = still[0] and not np.any(poses[:K - 1, 0] < srt[0, 3] / 1000.0)
    assert np.array_equal(got[still, :3], srt[still, :3])


@pytest.mark.parametrize("n,seed", [(5000, 1), (2, 2), (1, 3)])
def test_undistort_cv_against_independent_numpy(oracle, n, seed):
    rng = np.random.default_rng(seed)
    pts = np.c_[rng.uniform(-30, 30, (n, 3)), rng.uniform(0.0, 100.0, n)].astype(np.float32)
    omega, vel = rng.normal(0, 0.5, 3), rng.normal(0, 1.0, 3)
    end_R = Rotation.from_rotvec(rng.normal(0, 0.2, 3)).as_matrix()
    got = oracle.undistort_cv(pts, omega, vel, end_R)
    srt = pts[np.argsort(pts[:, 3], kind="stable")]
    ref = srt.copy()
    t = srt[:, 3].astype(np.float64) / 1000.0
    dt = t[-1] - t
    for j in range(1, n):                                   # the first point (in time) is skipped by the reference's loop bound
        ref[j, :3] = (_exp(omega, -dt[j]) @ srt[j, :3].astype(np.float64) - (end_R.T @ vel) * dt[j]).astype(np.float32)
    assert np.array_equal(got[:, 3], srt[:, 3])
    assert np.allclose(got[:, :3], ref[:, :3], rtol=0, atol=2e-5)
    assert np.array_equal(got[0, :3], srt[0, :3])
