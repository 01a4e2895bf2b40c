"""GPU parity of one registration pass (k_knn_ck + k_fit_reduce + k_reduce91) and of the full iterated update
against the oracle, through the C-ABI.  Tolerances (stated per SURVEY.md Appendix B):
  * neighbour lists: BIT-EXACT (same 5 points, same float32 squared distances, same order);
  * selected set: identical except at 1-ulp threshold flips (Jaccard >= 0.999);
  * H^T R^-1 H / H^T R^-1 z: relative 1e-9 when the selected sets are identical;
  * final pose of the update: |dp| <= 1e-6 m, |dtheta| <= 1e-7 rad.
"""
import numpy as np
import pytest

from conftest import make_state

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def reg(small_world):
    import lidar_imu_init_amd as lii
    hall, map_pts = small_world
    r = lii.Registrar(max_scan_points=150_000, max_map_points=400_000, filter_size_map=0.15)
    yield r
    r.close()


def _scan(small_world, sensor="tiny", seed=3):
    from harness import synth
    hall, _ = small_world
    R = synth.rot_zyx(0.03, -0.02, 0.4)
    p = np.array([0.8, -0.6, 0.1])
    return synth.make_scan(hall, sensor, R, p, noise=0.02, seed=seed), R, p


def _rel(a, b):
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


@pytest.mark.parametrize("sensor,imu_en", [("tiny", False), ("tiny", True), ("vlp16", True)])
def test_iterate_matches_oracle(reg, oracle, small_world, sensor, imu_en):
    import lidar_imu_init_amd as lii
    hall, map_pts = small_world
    scan, R, p = _scan(small_world, sensor)
    # perturbed start pose + a non-trivial extrinsic
    from harness import synth
    R_LI = synth.rot_zyx(0.01, 0.02, -0.015) if imu_en else np.eye(3)
    T_LI = np.array([0.03, -0.02, 0.05]) if imu_en else np.zeros(3)
    # body = LiDAR frame; world pose of the IMU frame chosen so that the LiDAR pose equals (R, p) + small error
    Rw = R @ synth.rot_zyx(0.004, -0.003, 0.005) @ R_LI.T
    pw = p + np.array([0.03, -0.02, 0.01]) - Rw @ T_LI
    st = make_state(oracle, Rw, pw, R_LI, T_LI)
    tree = oracle.Tree("oracle")
    tree.build(map_pts)
    ref = tree.iterate_once(scan, st, search=True, imu_en=imu_en, threads=4)
    reg.map_build(map_pts)
    reg.scan_upload(scan)
    n = reg.downsample_skip()
    assert n == len(scan)
    out = reg.iekf_iterate(lii.State(st), True, imu_en)
    nb, cnt, sel = reg.neighbors(n)
    assert np.array_equal(cnt, ref["nearest_n"])
    full = cnt == 5
    assert full.mean() > 0.9
    assert np.array_equal(nb[full], ref["nearest"][full]), "neighbour lists must be bit-exact"
    inter = np.logical_and(sel, ref["selected"]).sum()
    union = np.logical_or(sel, ref["selected"]).sum()
    assert inter / union >= 0.999
    if inter == union:
        assert int(out[90]) == int(ref["out91"][90])
        assert _rel(out[:90], ref["out91"][:90]) < 1e-9
    # a second, non-search pass at a moved pose must reuse the cached planes and the sticky selection
    st2 = oracle.state_boxplus(st, np.r_[0.001, -0.002, 0.0015, 0.01, -0.005, 0.004, np.zeros(18)])
    ref2 = tree.iterate_once(scan, st2, search=False, imu_en=imu_en, threads=4, selected=ref["selected"])
    out2 = reg.iekf_iterate(lii.State(st2), False, imu_en)
    _, _, sel2 = reg.neighbors(n)
    if inter == union and np.array_equal(sel2, ref2["selected"]):
        assert _rel(out2[:90], ref2["out91"][:90]) < 1e-9
    assert np.logical_xor(sel2, ref2["selected"]).sum() <= max(2, int(0.001 * n))


@pytest.mark.parametrize("imu_en,max_it", [(False, 4), (True, 5)])
def test_update_matches_oracle(reg, oracle, small_world, imu_en, max_it):
    import lidar_imu_init_amd as lii
    from harness import synth
    hall, map_pts = small_world
    scan, R, p = _scan(small_world, "vlp16", seed=11)
    st_true = make_state(oracle, R, p)
    st0 = oracle.state_boxplus(st_true, np.r_[0.006, -0.004, 0.008, 0.04, -0.03, 0.02, np.zeros(18)])
    tree = oracle.Tree("oracle")
    tree.build(map_pts)
    ref = tree.iekf_update(scan, st0, st0, max_iterations=max_it, imu_en=imu_en, threads=4)
    reg.map_build(map_pts)
    reg.scan_upload(scan)
    reg.downsample_skip()
    s = lii.State(st0)
    rep = reg.iekf_update(s, lii.State(st0), max_iterations=max_it, imu_en=imu_en)
    assert rep["iterations"] == ref["iters"]
    v, w = oracle.StateView(ref["state"]), s
    assert np.linalg.norm(v.pos_end - w.pos_end) <= 1e-6
    dR = v.rot_end.T @ w.rot_end
    assert np.linalg.norm(oracle.log_so3(dR)) <= 1e-7
    # the covariance of the weakly observable extrinsic block amplifies a 1-point selection flip (~3e-5 of H^T H)
    assert np.max(np.abs(v.cov - w.cov)) <= 1e-5 * max(1.0, np.max(np.abs(v.cov)))
    # and both recover the true LiDAR pose to the noise level (in LIO mode the state holds the IMU pose and the
    # extrinsic, so compose them)
    R_lidar = w.rot_end @ w.offset_R_L_I
    p_lidar = w.rot_end @ w.offset_T_L_I + w.pos_end
    assert np.linalg.norm(p_lidar - p) < 0.01
    assert np.linalg.norm(oracle.log_so3(R.T @ R_lidar)) < 0.002


def test_update_with_lio_regime_covariance(reg, oracle, small_world):
    """The covariance of a running LIO filter: pose block collapsed (1e-8) next to velocity / bias blocks of order 1 with
    strong cross-correlations, and a propagated state that differs from the current one in all 24 dimensions.  This is the
    regime where a gain evaluated as P21 - P21 G M P11 cancels catastrophically; all 24 states must follow the oracle's
    literal two-inversion algebra."""
    import lidar_imu_init_amd as lii
    hall, map_pts = small_world
    scan, R, p = _scan(small_world, "vlp16", seed=21)
    st_true = make_state(oracle, R, p)
    rng = np.random.default_rng(5)
    scale = np.sqrt(np.r_[np.full(6, 1e-8), np.full(6, 1e-4), np.full(3, 1.0), np.full(3, 1e-3), np.full(3, 1e-2), np.full(3, 1e-5)])
    A = rng.normal(0, 1, (24, 24))
    C = A @ A.T / 24 + np.eye(24)
    C = C / np.sqrt(np.outer(np.diag(C), np.diag(C)))          # a correlation matrix with off-diagonals up to ~0.4
    P = C * np.outer(scale, scale)
    for imu_en in (True, False):
        prop = lii.State(oracle.state_boxplus(st_true, np.r_[2e-4, -1e-4, 2e-4, 2e-3, -1e-3, 1e-3, np.zeros(18)]))
        prop.cov[:] = P
        cur = lii.State(oracle.state_boxplus(prop.pod, np.r_[rng.normal(0, 1e-4, 6), rng.normal(0, 1e-3, 6), rng.normal(0, 1e-2, 12)]))
        tree = oracle.Tree("oracle")
        tree.build(map_pts)
        ref = tree.iekf_update(scan, cur.pod, prop.pod, max_iterations=5, imu_en=imu_en, threads=4)
        reg.map_build(map_pts)
        reg.scan_upload(scan)
        reg.downsample_skip()
        s = cur.copy()
        rep = reg.iekf_update(s, prop, max_iterations=5, imu_en=imu_en)
        assert rep["iterations"] == ref["iters"]
        v = oracle.StateView(ref["state"])
        d = oracle.state_boxminus(s.pod, ref["state"])
        assert np.max(np.abs(d[:12])) < 1e-7, d[:12]          # pose + extrinsic
        assert np.max(np.abs(d[12:])) < 1e-5, d[12:]          # velocity, biases, gravity (prior std 1 .. 3e-3)
        assert np.max(np.abs(v.cov - s.cov)) <= 1e-6 * np.max(np.abs(v.cov))

def test_elimination_with_row_exchanges_when_the_pivot_vanishes(reg, oracle, small_world):
    """The device solve eliminates [I + P11 G | P[:12, :]] without row exchanges while the elimination's element growth stays below
    2^8, and repeats it with threshold pivoting otherwise (lii_iekf.hip: gj12_loop / gj12_pivoting).  A prior whose pose block is
    eps I + s v v^T with v chosen against the scene's G makes the first pivot 1 + (P11 G)_00 vanish (1e-7): the pivot-free form
    would divide by it.  The update must report the routine with exchanges (lii_last_solve_info) and land where the oracle's
    two-inversion algebra (Eigen's partial-pivoting LU, src/laserMapping.cpp:1081-1085) lands; an ordinary prior reports none."""
    import lidar_imu_init_amd as lii
    hall, map_pts = small_world
    scan, R, p = _scan(small_world, "vlp16", seed=11)
    st_true = make_state(oracle, R, p)
    st0 = oracle.state_boxplus(st_true, np.r_[0.002, -0.001, 0.002, 0.01, -0.01, 0.005, np.zeros(18)])
    reg.map_build(map_pts)
    reg.scan_upload(scan)
    reg.downsample_skip()
    out = reg.iekf_iterate(lii.State(st0), True, True)
    G = np.zeros((12, 12))
    G[np.triu_indices(12)] = out[:78]
    G = G + np.triu(G, 1).T
    u = G[:, 0]
    j = 1 + int(np.argmax(np.abs(u[1:])))
    v = np.zeros(12)
    v[0], v[j] = 1.0, -2.0 * u[0] / u[j]                      # v . u = -u0 < 0
    eps, delta = 1e-10, 1e-7
    s_ = (1.0 - delta + eps * u[0]) / u[0]                     # 1 + eps u0 + s (v0)(v . u) = delta
    P11 = eps * np.eye(12) + s_ * np.outer(v, v)
    assert abs(1.0 + (P11 @ G)[0, 0]) < 1e-5 and np.max(np.abs((np.eye(12) + P11 @ G)[1:, 0])) > 1e-2
    prop = lii.State(st0)
    prop.cov[:] = 0
    prop.cov[:12, :12] = P11
    prop.cov[12:, 12:] = 1e-4 * np.eye(12)
    tree = oracle.Tree("oracle")
    tree.build(map_pts)
    ref = tree.iekf_update(scan, prop.pod, prop.pod, max_iterations=4, imu_en=True, threads=4)
    s = prop.copy()
    rep = reg.iekf_update(s, prop, max_iterations=4, imu_en=True)
    assert reg.last_solve_info() >= 1
    assert rep["iterations"] == ref["iters"]
    d = oracle.state_boxminus(s.pod, ref["state"])
    assert np.max(np.abs(d[:12])) < 1e-7, d[:12]
    assert np.max(np.abs(d[12:])) < 1e-6, d[12:]
    v_ = oracle.StateView(ref["state"])
    assert np.max(np.abs(v_.cov - s.cov)) <= 1e-6 * np.max(np.abs(v_.cov))
    # an ordinary prior: no pass needs the exchanges
    s2 = lii.State(st0)
    reg.iekf_update(s2, lii.State(st0), max_iterations=4, imu_en=True)
    assert reg.last_solve_info() == 0


def test_sparse_and_empty_neighbourhoods(reg, oracle):
    """Frontier behaviour: queries with fewer than 5 neighbours within sqrt(5) m, phase-2 ring search."""
    import lidar_imu_init_amd as lii
    rng = np.random.default_rng(5)
    # a sparse map: 4000 points scattered in a 60 m cube -> most 5-NN radii exceed the 3x3x3 cell block
    map_pts = rng.uniform(-30, 30, (4000, 3)).astype(np.float32)
    q = rng.uniform(-34, 34, (5000, 3)).astype(np.float32)
    scan = np.c_[q, np.zeros(len(q), np.float32)]
    tree = oracle.Tree("oracle")
    tree.build(map_pts)
    st = oracle.state_init()
    ref = tree.iterate_once(scan, st, search=True, imu_en=False, threads=4)
    reg.map_build(map_pts)
    reg.scan_upload(scan)
    n = reg.downsample_skip()
    reg.iekf_iterate(lii.State(st), True, False)
    nb, cnt, sel = reg.neighbors(n)
    assert np.array_equal(cnt, ref["nearest_n"])
    assert (cnt < 5).any() and (cnt == 5).any()
    for k in range(5):
        m = cnt > k
        assert np.array_equal(nb[m, k], ref["nearest"][m, k])


def test_communicator_path_world_size_one(oracle, small_world):
    """With a communicator attached the loop runs as separate final-sum / RCCL all-reduce / solve launches; at world size 1
    the all-reduce is the identity, so the result must be BIT-identical to the fused single-GPU loop."""
    import lidar_imu_init_amd as lii
    hall, map_pts = small_world
    scan, R, p = _scan(small_world, "vlp16", seed=31)
    st_true = make_state(oracle, R, p)
    st0 = oracle.state_boxplus(st_true, np.r_[0.003, -0.002, 0.004, 0.03, -0.02, 0.01, np.zeros(18)])
    out = []
    for comm in (False, True):
        r = lii.Registrar(max_scan_points=40_000, max_map_points=400_000, filter_size_map=0.15)
        r.map_build(map_pts)
        if comm:
            r.comm_init(1, 0, r.comm_unique_id())
        r.scan_upload(scan)
        r.downsample_skip()
        s = lii.State(st0)
        rep = r.iekf_update(s, lii.State(st0), max_iterations=5, imu_en=True)
        sums = r.iekf_iterate(s, True, True)  # the host-driven single pass goes through the all-reduce as well
        out.append((s.pod.copy(), rep, sums))
        r.close()
    assert np.array_equal(out[0][0], out[1][0])
    assert out[0][1]["iterations"] == out[1][1]["iterations"] and np.array_equal(out[0][1]["normal_eq"], out[1][1]["normal_eq"])
    assert np.array_equal(out[0][2], out[1][2])

