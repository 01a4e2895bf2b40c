"""GPU parity of the per-scan point kernels through the C-ABI:
  * lii_undistort_imu / lii_undistort_cv vs the oracle's restatement of src/IMU_Processing.hpp:390-414 / :246-266
    — tolerance: <= 2 ulp(float32) per coordinate (device sin/cos vs glibc differ in the last double bit);
  * lii_downsample vs the PCL VoxelGrid restatement — BIT-EXACT (same float32 sums in the same order);
  * edge cases: empty scan, single point, all points in one voxel, non-finite points, the int32-overflow identity path.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def reg():
    import lidar_imu_init_amd as lii
    r = lii.Registrar(max_scan_points=300_000, max_map_points=10_000, filter_size_map=0.2)
    yield r
    r.close()


def _ulp_diff(a, b):
    a = np.ascontiguousarray(a, np.float32).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(b, np.float32).view(np.int32).astype(np.int64)
    a = np.where(a < 0, -(a & 0x7FFFFFFF), a)
    b = np.where(b < 0, -(b & 0x7FFFFFFF), b)
    return np.abs(a - b)


def _pose_table(rng, K, sweep_s=0.1, t0=0.0):
    from lidar_imu_init_amd import pose6d_array
    from harness import synth
    T = pose6d_array(K)
    times = np.sort(np.r_[t0, rng.uniform(0.002, sweep_s, K - 1)])
    R = np.eye(3)
    p = np.zeros(3)
    v = rng.normal(0, 0.5, 3)
    for k in range(K):
        T[k, 0] = times[k]
        T[k, 1:4] = rng.normal(0, 1.0, 3)       # acc
        T[k, 4:7] = rng.normal(0, 0.8, 3)       # gyr
        T[k, 7:10] = v + rng.normal(0, 0.05, 3)
        T[k, 10:13] = p + rng.normal(0, 0.02, 3)
        T[k, 13:22] = (R @ synth.rot_zyx(*rng.normal(0, 0.02, 3))).reshape(-1)
    return T


@pytest.mark.parametrize("n,K,tmin", [(50_000, 12, 0.0), (4097, 3, 0.7), (1, 5, 3.0), (257, 2, 0.0)])
def test_undistort_imu_matches_oracle(reg, oracle, n, K, tmin):
    from harness import synth
    rng = np.random.default_rng(100 + n)
    pts = np.c_[rng.uniform(-40, 40, (n, 3)), rng.uniform(tmin, 100.0, n)].astype(np.float32)
    if n > 10:
        pts[rng.choice(n, 5, replace=False), 3] = np.float32(tmin)  # several time-earliest points (quirk A3)
        pts[3, 3] = 100.0
    T = _pose_table(rng, K)
    end_R = synth.rot_zyx(0.02, -0.03, 0.1)
    end_p = np.array([0.3, -0.2, 0.05])
    R_LI = synth.rot_zyx(-0.016, -0.0057, 1.538)
    T_LI = np.array([-0.02, 0.02, 0.17])
    ref = oracle.undistort_imu(pts, T, end_R, end_p, R_LI, T_LI)  # stable time sort + de-skew
    reg.scan_upload(pts)
    reg.undistort_imu(T, end_R, end_p, R_LI, T_LI)
    got = reg.scan_download(0)
    order = np.argsort(pts[:, 3], kind="stable")  # the GPU keeps the input order; the oracle returns it time-sorted
    assert np.array_equal(got[order, 3], ref[:, 3])
    assert _ulp_diff(got[order, :3], ref[:, :3]).max() <= 2
    # points at or before every pose-table time must be untouched
    untouched = pts[:, 3] / 1000.0 <= T[0, 0]
    assert np.array_equal(got[untouched, :3], pts[untouched, :3])


@pytest.mark.parametrize("n", [30_000, 1, 2, 1000])
def test_undistort_cv_matches_oracle(reg, oracle, n):
    from harness import synth
    rng = np.random.default_rng(7 + n)
    pts = np.c_[rng.uniform(-30, 30, (n, 3)), rng.uniform(0.0, 100.0, n)].astype(np.float32)
    if n > 10:
        pts[[5, 9], 3] = pts[:, 3].min()  # tie for "first point": the lowest index is skipped
    omega = np.array([0.4, -0.3, 0.9])
    vel = np.array([1.2, -0.4, 0.1])
    end_R = synth.rot_zyx(0.1, -0.05, 0.7)
    ref = oracle.undistort_cv(pts, omega, vel, end_R)
    reg.scan_upload(pts)
    reg.undistort_cv(omega, vel, end_R)
    got = reg.scan_download(0)
    order = np.argsort(pts[:, 3], kind="stable")
    assert _ulp_diff(got[order, :3], ref[:, :3]).max() <= 2
    assert np.array_equal(got[order[0], :3], pts[order[0], :3])  # the time-earliest point is never compensated


@pytest.mark.parametrize("n,leaf,spread", [(120_000, 0.05, 30.0), (50_000, 0.5, 20.0), (5000, 5.0, 1.0), (1, 0.05, 1.0),
                                           # the sample sort's regimes: one-workgroup path, smallest bucket count, every
                                           # point in ONE voxel (equal keys must not overload a bucket), a large scan
                                           (512, 0.5, 5.0), (513, 0.5, 5.0), (2049, 0.2, 3.0), (20_000, 50.0, 1.0),
                                           (149_000, 0.1, 40.0)])
def test_voxel_grid_matches_oracle(reg, oracle, n, leaf, spread):
    rng = np.random.default_rng(n)
    pts = np.c_[rng.uniform(-spread, spread, (n, 2)), rng.uniform(-2, 4, n), rng.uniform(0, 100, n)].astype(np.float32)
    if n > 100:
        pts[:50, :3] = pts[50:100, :3] + np.float32(0.001)  # several points per voxel
    ref, filtered = oracle.voxel_grid(pts, leaf)
    assert filtered
    reg.scan_upload(pts)
    nd, f = reg.downsample(leaf)
    assert f and nd == len(ref)
    got = reg.scan_download(1)
    assert np.array_equal(got, ref), "voxel centroids must be bit-exact (same float32 summation order)"


def test_voxel_grid_edge_cases(reg, oracle):
    # empty scan
    reg.scan_upload(np.zeros((0, 4), np.float32))
    nd, f = reg.downsample(0.05)
    assert nd == 0
    # non-finite points are dropped
    pts = np.array([[0, 0, 0, 1], [np.nan, 0, 0, 2], [0.01, 0, 0, 3], [np.inf, 1, 1, 4], [3, 3, 3, 5]], np.float32)
    ref, _ = oracle.voxel_grid(pts, 0.05)
    reg.scan_upload(pts)
    nd, f = reg.downsample(0.05)
    assert nd == len(ref) == 2
    assert np.array_equal(reg.scan_download(1), ref)
    # PCL's int32 index-overflow guard: a 200 m cube at 5 cm leaves -> identity copy, order preserved
    rng = np.random.default_rng(0)
    big = np.c_[rng.uniform(-100, 100, (1000, 3)), rng.uniform(0, 100, 1000)].astype(np.float32)
    ref, filtered = oracle.voxel_grid(big, 0.05)
    assert not filtered and np.array_equal(ref, big)
    reg.scan_upload(big)
    nd, f = reg.downsample(0.05)
    assert (not f) and nd == 1000
    assert np.array_equal(reg.scan_download(1), big)


def test_voxel_grid_sort_survives_adversarial_sampling(reg, oracle):
    """The voxel filter's sample sort draws one jittered sample per stratum of the input (k_voxel_keys).  Here every sampled
    position holds a point of the lowest voxels and everything else lies above them: all splitters collapse to the bottom, one
    bucket receives ~the whole scan, and the oversized-group path (ranking out of global memory) has to produce the same
    bit-exact result."""
    n, leaf = 20_000, 0.1
    rng = np.random.default_rng(5)
    pts = np.c_[rng.uniform(5, 25, (n, 2)), rng.uniform(1, 4, n), rng.uniform(0, 100, n)].astype(np.float32)
    # the library's sampling plan (lii_vsort.hip: voxel_sort_plan / k_voxel_keys), restated
    want, B = (n + 63) // 64, 8
    while B < want and B < 2048:
        B *= 2
    S = 2 * B
    W = (n + S - 1) // S
    strata = (n + W - 1) // W
    M32 = 0xFFFFFFFF
    for j in range(strata):
        h = (j * 2654435761) & M32
        h ^= h >> 15
        h = (h * 2246822519) & M32
        h ^= h >> 13
        lo = j * W
        pos = lo + h % min(W, n - lo)
        pts[pos, :3] = np.float32([0.01 + 1e-4 * (j % 7), 0.01, 0.01])  # the lowest corner of the grid
    ref, filtered = oracle.voxel_grid(pts, leaf)
    assert filtered
    reg.scan_upload(pts)
    nd, f = reg.downsample(leaf)
    assert f and nd == len(ref)
    assert np.array_equal(reg.scan_download(1), ref)


@pytest.mark.parametrize("n", [700, 1000, 4095, 4096, 8191, 8192, 8193, 16385, 65536, 131072, 131073, 200000])
def test_voxel_grid_sizes_around_the_sort_thresholds(reg, oracle, n):
    """Bucket counts, sample counts and the group size of the voxel filter's sort change at these sizes (lii_vsort.hip:
    voxel_sort_plan, the 131 k switch to one bucket per workgroup); half of the points share voxels with others."""
    rng = np.random.default_rng(n)
    pts = np.c_[rng.uniform(-20, 20, (n, 2)), rng.uniform(-1, 3, n), rng.uniform(0, 100, n)].astype(np.float32)
    pts[n // 2:, :3] = pts[rng.integers(0, n // 2, n - n // 2), :3] + rng.uniform(0, 0.02, (n - n // 2, 3)).astype(np.float32)
    ref, filtered = oracle.voxel_grid(pts, 0.1)
    reg.scan_upload(pts)
    nd, f = reg.downsample(0.1)
    assert f == filtered and nd == len(ref)
    assert np.array_equal(reg.scan_download(1), ref)


def test_next_scan_travels_while_the_current_one_is_in_use(reg):
    """lii_scan_upload_next / lii_scan_advance (the reference's one-deep scan queue, src/laserMapping.cpp:331-366): the scan that
    arrives through the second buffer is the scan lii_scan_upload would have delivered - from pageable memory (staged), from
    pinned memory (read by the copy engine), and from 48-byte PointXYZINormal records - and the current scan is untouched
    until the swap."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")  # the runtime the library itself uses: pinned host memory for the direct path
    rng = np.random.default_rng(11)
    a = rng.normal(size=(30_000, 4)).astype(np.float32)
    b = rng.normal(size=(41_000, 4)).astype(np.float32)
    c12 = rng.normal(size=(5_000, 12)).astype(np.float32)
    reg.scan_upload(a)
    reg.scan_upload_next(b)                      # pageable
    assert np.array_equal(reg.scan_download(0), a)
    reg.scan_advance()
    assert np.array_equal(reg.scan_download(0), b)
    ptr = C.c_void_p()
    assert hip.hipHostMalloc(C.byref(ptr), C.c_size_t(a.nbytes), C.c_uint(0)) == 0
    pinned = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), shape=(a.size,)).reshape(a.shape)
    pinned[:] = a
    reg.scan_upload_next(pinned)                 # pinned: direct
    assert np.array_equal(reg.scan_download(0), b)
    reg.scan_advance()
    assert np.array_equal(reg.scan_download(0), a)
    reg.scan_upload_next(c12)                    # strided records
    reg.scan_upload_next(b)                      # replaces a scan that was never advanced to
    reg.scan_advance()
    assert np.array_equal(reg.scan_download(0), b)
    reg.scan_upload_next(c12)
    reg.scan_advance()
    got = reg.scan_download(0)
    assert np.array_equal(got[:, :3], c12[:, :3]) and np.array_equal(got[:, 3], c12[:, 9])
    with pytest.raises(Exception):
        reg.scan_advance()                       # nothing under way
    reg.scan_upload_next(np.zeros((0, 4), np.float32))
    reg.scan_advance()
    assert len(reg.scan_download(0)) == 0
    assert hip.hipHostFree(ptr) == 0


def test_pipelined_hand_over_gives_the_same_stream(oracle):
    """A short stream processed twice - serial lii_scan_upload per scan, and with every next scan travelling through
    lii_scan_upload_next while its predecessor is registered and inserted into the map - ends in the same states (bit for bit) and
    the same map (as a set)."""
    import bench
    import lidar_imu_init_amd as lii
    wl = bench.build_workload("vlp16", 4)
    states0, tables = bench.start_states(wl)
    scans = [np.ascontiguousarray(s, np.float32) for s in wl["scans"]]

    def run(pipelined):
        reg = lii.Registrar(max_scan_points=40_000, max_map_points=400_000, filter_size_map=wl["fs_map"])
        reg.map_build(wl["map"])
        out = []
        if pipelined:
            reg.scan_upload_next(scans[0])
            reg.scan_advance()
        for k in range(len(scans)):
            if pipelined:
                if k + 1 < len(scans):
                    reg.scan_upload_next(scans[k + 1])
            else:
                reg.scan_upload(scans[k])
            st = states0[k].copy()
            rep = reg.scan_register(st, states0[k], imu_poses=tables[k], leaf=wl["fs_surf"], max_iterations=wl["max_it"], imu_en=True)
            reg.map_incremental(st)
            out.append((st.pod.copy(), rep["iterations"], rep["searches"]))
            if pipelined and k + 1 < len(scans):
                reg.scan_advance()
        m = np.unique(reg.map_download(), axis=0)
        reg.close()
        return out, m

    a, ma = run(False)
    b, mb = run(True)
    for x, y in zip(a, b):
        assert x[1:] == y[1:] and np.array_equal(x[0], y[0])
    assert ma.shape == mb.shape and np.array_equal(ma, mb)
