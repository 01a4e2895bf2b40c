"""The configurations bench.py times, gated against the oracle END TO END: lii_scan_register (IMU back-propagation de-skew ->
voxel grid leaf 0.05 -> device-driven iterated update, LIO mode, max_iteration 5) on the north-star stream (100 000 pts/scan),
OS1-128 (131 072), its cut_frame_num = 3 sub-frames (~43.7 k, BASELINE.json configs[3]) and the dense 500 k scan, all against
the 1 M-point map - the same scans, map, start states and pose tables bench.py builds (bench.build_workload) - versus the
oracle's undistort_imu -> voxel_grid -> Tree.iekf_update on the host.

Tolerances (tests/test_gpu_register.py): iterations and k-NN passes equal; effect_num within the 1-ulp
threshold flips (<= 0.01 % of the points); the down-sampled cloud bit-identical (in the reference's order); final state:
  * up to 131 072 points per scan: |dp| <= 1e-6 m, |dtheta| <= 1e-7 rad, pose + extrinsic <= 1e-7, other states <= 1e-5
    (boxminus), covariance <= 5e-4 of its largest entry - and, derived rather than asserted: within 1e-4 (element-wise, relative to
    sqrt(P_ii P_jj)) of the EXACT posterior (P^-1 + H^T R^-1 H)^-1 computed in 60-digit arithmetic, and never farther from it than
    the reference's own algebra (the oracle) is;
  * ~500 k points: |dtheta| <= 1e-6 rad, covariance <= 2e-3.  In LIO mode a correction is split between the IMU attitude and
    the extrinsic rotation by the (unit) prior alone, the normal matrix P^-1 + H^T R^-1 H has cond ~ 2e11 at 340 k effective
    points, and BOTH algebras - the reference's two 24 x 24 inversions restated by the oracle, and the device's single 12-step
    elimination - sit 4e-8 .. 8e-8 rad from the exact (80-bit) solution of the same normal equations: 1e-7 rad is below the
    conditioning noise of the reference's own arithmetic at this size (measured GPU vs oracle: 2.2e-7 rad, 1.2e-7 m; the
    composed LiDAR pose R_end R_LI, R_end T_LI + p_end carries the same noise: 1.3e-7 rad)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def world():
    import bench
    import lidar_imu_init_amd as lii
    cache = {}
    wl = bench.build_workload("stream100k", 2, map_cache=cache)
    reg = lii.Registrar(max_scan_points=520_000, max_map_points=1_100_000, filter_size_map=wl["fs_map"])
    reg.map_build(wl["map"])
    from oracle import oracle as O
    tree = O.Tree("oracle")
    tree.build(wl["map"])
    yield cache, reg, tree
    reg.close()


@pytest.mark.parametrize("workload,n_scans", [("stream100k", 2), ("os1_128", 1), ("os1_128_cut3", 3), ("dense500k", 1)])
def test_scan_register_matches_oracle_at_bench_size(world, oracle, workload, n_scans):
    cache, reg, tree = world
    _check_workload(cache, reg, tree, oracle, workload, n_scans)


def test_scan_register_matches_oracle_on_the_vlp16_configuration(oracle):
    """BASELINE.json configs[1] at ITS size: ~30 k points per scan against the 300 k-point map (every other headline configuration
    registers against the 1 M-point map of the fixture above) - VERDICT r4 weak 2."""
    import bench
    import lidar_imu_init_amd as lii
    cache = {}
    wl = bench.build_workload("vlp16", 2, map_cache=cache)
    assert 250_000 <= len(wl["map"]) <= 350_000 and 25_000 <= len(wl["scans"][0]) <= 35_000
    reg = lii.Registrar(max_scan_points=40_000, max_map_points=400_000, filter_size_map=wl["fs_map"])
    try:
        reg.map_build(wl["map"])
        tree = oracle.Tree("oracle")
        tree.build(wl["map"])
        _check_workload(cache, reg, tree, oracle, "vlp16", 2)
    finally:
        reg.close()


def _check_workload(cache, reg, tree, oracle, workload, n_scans):
    import bench
    wl = bench.build_workload(workload, n_scans, map_cache=cache)
    assert wl["fs_surf"] == 0.05 and wl["max_it"] == 5  # read from harness/config + harness/launch (reference format)
    states0, tables = bench.start_states(wl)
    for j, scan in enumerate(wl["scans"]):
        ref = bench.oracle_scan_register(oracle, tree, scan, states0[j], tables[j], wl["fs_surf"], wl["max_it"], threads=8)
        st = states0[j].copy()
        rep = reg.scan_register(st, states0[j], imu_poses=tables[j], leaf=wl["fs_surf"], max_iterations=wl["max_it"], imu_en=True,
                                scan_dev=reg.device_scan(scan), scan_sorted=True)  # (the bench's streams are time-sorted)
        body = reg.scan_download(1)
        assert len(body) == ref["n_down"]
        par = bench.parity_against_oracle(oracle, ref, st.pod, rep, prior_cov=states0[j].cov)
        print(workload, j, len(scan), "->", len(body), rep["iterations"], rep["searches"], rep["effect_num"], par)
        assert par["iters_equal"] and par["searches_equal"], (rep, ref["iters"], ref["logs"][:, :2])
        big = len(scan) > 200_000
        assert par["dp"] <= 1e-6 and par["dtheta"] <= (1e-6 if big else 1e-7)
        assert par["dp_lidar"] <= 1e-6 and par["dtheta_lidar"] <= (1e-6 if big else 1e-7)
        assert par["dstate_pose_ext"] <= (1e-6 if big else 1e-7) and par["dstate_rest"] <= 1e-5
        assert par["dcov_rel"] <= (2e-3 if big else 5e-4)
        # The covariance bound, derived: both posteriors against the EXACT posterior (P^-1 + G (+) 0)^-1 of the same normal
        # equations (60-digit arithmetic), element-wise relative to sqrt(P_ii P_jj).  Measured on these scans: the device's
        # 12-step elimination + symmetrisation sits 2e-6 .. 2e-5 from it, the reference's two 24 x 24 inversions (the oracle)
        # 2e-5 .. 1.1e-3 - the GPU-vs-oracle difference above IS the oracle's own arithmetic noise.
        assert par["dcov_gpu_exact"] <= (2e-4 if big else 1e-4), par
        assert par["dcov_gpu_exact"] <= par["dcov_oracle_exact"] + 1e-6, par  # never farther from the truth than the reference algebra
        assert par["dcov_gpu_oracle"] <= 2.0 * par["dcov_oracle_exact"] + 1e-5, par
        assert par["effect_diff"] <= max(2, int(1e-4 * len(body)))
        # and the registration really converged onto the scene (ground truth of the synthetic stream): the LiDAR pose - in LIO
        # mode half of the start error of the IMU attitude stays in R_end and the other half moves into the extrinsic
        R, p = wl["poses"][j]
        R_lidar = st.rot_end @ st.offset_R_L_I
        p_lidar = st.rot_end @ st.offset_T_L_I + st.pos_end
        assert np.linalg.norm(p_lidar - p) < 0.03  # (a third of a sweep constrains the pose less: ~2 cm)
        assert np.linalg.norm(oracle.log_so3(R.T @ R_lidar)) < 2e-3


def test_downsampled_cloud_is_bit_identical_at_bench_size(world, oracle):
    """The de-skewed, voxel-filtered cloud the update starts from - 100 k points through IMU back-propagation and the leaf-0.05
    grid - equals the oracle's bit for bit, in the reference's (PCL index) order, although the device keeps it in the order of the voxels' first points."""
    import bench
    cache, reg, tree = world
    wl = bench.build_workload("stream100k", 1, map_cache=cache)
    states0, tables = bench.start_states(wl)
    s0 = states0[0]
    und = oracle.undistort_imu(wl["scans"][0], tables[0], s0.rot_end, s0.pos_end, s0.offset_R_L_I, s0.offset_T_L_I)
    ref, filtered = oracle.voxel_grid(und, wl["fs_surf"])
    assert filtered
    reg.scan_upload(wl["scans"][0])
    reg.undistort_imu(tables[0], s0.rot_end, s0.pos_end, s0.offset_R_L_I, s0.offset_T_L_I)
    got_und = reg.scan_download(0)
    nd, f = reg.downsample(wl["fs_surf"])
    got = reg.scan_download(1)
    assert f and nd == len(ref)
    if np.array_equal(got_und, und):  # device sin/cos agree with glibc on this table (<= 2 ulp otherwise, test_gpu_scan_ops.py)
        assert np.array_equal(got, ref)
    else:
        assert np.max(np.abs(got - ref)) <= 4e-6


@pytest.mark.parametrize("workload", ["stream100k", "dense500k"])
def test_registration_is_bit_identical_run_to_run(world, oracle, workload):
    """behind a search pass the completion workgroups of k_fit_reduce rewrite the flagged queries' lists and counts
    WHILE the workgroups of the cloud of the same launch decide from those counts whether a point is theirs.  Ownership now rests on
    flags that say "not mine" before (kNeedy) and after (kDone) the completion: the normal equations - and with them every bit of
    the result - must not depend on which workgroup ran first.  dense500k is the launch with ~1 300 workgroups (more than the chip
    holds at once) the finding was about; five runs from the same state must agree to the last bit, the neighbour lists included."""
    import bench
    cache, reg, tree = world
    wl = bench.build_workload(workload, 1, map_cache=cache)
    states0, tables = bench.start_states(wl)
    dev = reg.device_scan(wl["scans"][0])
    first = None
    for run in range(5):
        st = states0[0].copy()
        rep = reg.scan_register(st, states0[0], imu_poses=tables[0], leaf=wl["fs_surf"], max_iterations=wl["max_it"], imu_en=True,
                                scan_dev=dev, scan_sorted=True)
        nb, cnt, sel = reg.neighbors(len(reg.scan_download(1)))
        got = (np.array(st.pod).tobytes(), rep["iterations"], rep["searches"], rep["effect_num"], nb.tobytes(), cnt.tobytes(), sel.tobytes())
        if first is None:
            first = got
            assert cnt.max() <= 5  # the completion flags never leave the library
        else:
            for a, b, what in zip(first, got, ("state", "iterations", "searches", "effect_num", "neighbours", "counts", "selected")):
                assert a == b, f"run {run}: {what} differs from run 0"
