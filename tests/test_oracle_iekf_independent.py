"""One iteration of the iterated Kalman update (reference src/laserMapping.cpp:973-1087) written INDEPENDENTLY in numpy / scipy
and compared with oracle/orc_iekf.hpp on small scans: pointBodyToWorld, plane fit (numpy lstsq instead of the restated QR),
residual + selection gate, Jacobian rows, and the LITERAL gain of the reference,
    K = (H^T R^-1 H (+) 0 + P^-1)^-1 [:, :12] H^T R^-1        (24 x m),   solution = K z + vec - K H vec[:12],
with numpy's LAPACK inverses, then boxplus with scipy rotations.  Only the neighbour lists are taken from the oracle (its k-d
tree is pinned bit for bit against the unmodified reference ikd-Tree in tests/test_oracle_core.py)."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation

from conftest import make_state


def numpy_iteration(body, nearest, nn, st, prop, imu_en, rinv=1000.0):
    R, p = st[0:9].reshape(3, 3), st[9:12]
    RLI, TLI = st[12:21].reshape(3, 3), st[21:24]
    P = st[36:].reshape(24, 24)
    rows, zs, sel = [], [], np.zeros(len(body), bool)
    for i, pb in enumerate(body[:, :3].astype(np.float64)):
        if nn[i] < 5:
            continue
        pw = (R @ (RLI @ pb + TLI) + p).astype(np.float32)              # pointBodyToWorld :209-220 (float result)
        A = nearest[i].astype(np.float64)
        if ((A - pw.astype(np.float64)) ** 2).sum(1).max() > 5.0 + 1e-3:
            continue
        n = np.linalg.lstsq(A, -np.ones(5), rcond=None)[0]                # esti_plane :236-269
        nn_ = np.linalg.norm(n)
        pabcd = np.r_[n / nn_, 1.0 / nn_]
        if np.any(np.abs(A @ pabcd[:3] + pabcd[3]) > 0.1):
            continue
        pd2 = np.float32(pabcd[0] * pw[0] + pabcd[1] * pw[1] + pabcd[2] * pw[2] + pabcd[3])
        s = np.float32(1 - 0.9 * abs(pd2) / np.sqrt(np.linalg.norm(pb)))  # :1001
        if not s > 0.9:
            continue
        sel[i] = True
        nv = pabcd[:3].astype(np.float32).astype(np.float64)              # normvec is stored as float (:1004-1006)
        pi = RLI @ pb + TLI
        C = R.T @ nv
        row = np.zeros(12)
        row[0:3] = np.cross(pi, C)                                       # [p_I]x R^T n
        row[3:6] = nv
        if imu_en:
            row[6:9] = np.cross(pb, RLI.T @ C)
            row[9:12] = C
        rows.append(row)
        zs.append(-float(pd2))
    H = np.array(rows)
    z = np.array(zs)
    G = np.zeros((24, 24))
    G[:12, :12] = H.T @ H * rinv
    K1 = np.linalg.inv(G + np.linalg.inv(P))
    K = K1[:, :12] @ H.T * rinv                                           # 24 x m, literally
    vec = np.zeros(24)
    Rp, RLIp = prop[0:9].reshape(3, 3), prop[12:21].reshape(3, 3)
    vec[0:3] = Rotation.from_matrix(R.T @ Rp).as_rotvec()
    vec[3:6] = prop[9:12] - p
    vec[6:9] = Rotation.from_matrix(RLI.T @ RLIp).as_rotvec()
    vec[9:12] = prop[21:24] - TLI
    vec[12:24] = prop[24:36] - st[24:36]
    sol = K @ z + vec - (K @ H) @ vec[:12]
    new = st.copy()
    new[0:9] = (R @ Rotation.from_rotvec(sol[0:3]).as_matrix()).reshape(-1)
    new[9:12] = p + sol[3:6]
    new[12:21] = (RLI @ Rotation.from_rotvec(sol[6:9]).as_matrix()).reshape(-1)
    new[21:24] = TLI + sol[9:12]
    new[24:36] = st[24:36] + sol[12:24]
    return dict(sel=sel, HTH=H.T @ H * rinv, HTz=H.T @ z * rinv, sol=sol, state=new, m=len(z))


@pytest.mark.parametrize("imu_en", [False, True])
def test_one_iteration_matches_independent_numpy(oracle, small_world, imu_en):
    from harness import synth
    hall, map_pts = small_world
    R = synth.rot_zyx(0.03, -0.02, 0.4)
    p = np.array([0.8, -0.6, 0.1])
    scan = synth.make_scan(hall, "tiny", R, p, noise=0.02, seed=3)
    R_LI = synth.rot_zyx(0.01, 0.02, -0.015) if imu_en else np.eye(3)
    T_LI = np.array([0.03, -0.02, 0.05]) if imu_en else np.zeros(3)
    Rw = R @ synth.rot_zyx(0.004, -0.003, 0.005) @ R_LI.T
    pw = p + np.array([0.03, -0.02, 0.01]) - Rw @ T_LI
    prop = make_state(oracle, Rw, pw, R_LI, T_LI)
    rng = np.random.default_rng(2)
    A = rng.normal(0, 1, (24, 24))
    Cm = A @ A.T / 24 + np.eye(24)
    scale = np.sqrt(np.r_[np.full(6, 1e-4), np.full(6, 1e-4), np.full(12, 1e-3)])
    prop[36:] = (Cm / np.sqrt(np.outer(np.diag(Cm), np.diag(Cm))) * np.outer(scale, scale)).reshape(-1)
    cur = oracle.state_boxplus(prop, np.r_[rng.normal(0, 2e-4, 6), rng.normal(0, 1e-4, 6) * imu_en, rng.normal(0, 1e-3, 12)])
    tree = oracle.Tree("oracle")
    tree.build(map_pts)
    ref = tree.iekf_update(scan, cur, prop, max_iterations=1, imu_en=imu_en, threads=2, literal_gain=True)
    mine = numpy_iteration(scan, ref["nearest"], ref["nearest_n"], cur, prop, imu_en)
    lg = ref["logs"][0]
    G = np.zeros((12, 12))
    G[np.triu_indices(12)] = lg[2:80]
    G = G + G.T - np.diag(np.diag(G))
    flips = int(np.logical_xor(mine["sel"], ref["selected"].astype(bool)).sum())
    assert flips <= 1 and abs(mine["m"] - int(lg[1])) <= 1          # the gate is a float32 comparison: a 1-ulp flip at most
    if flips == 0:
        assert np.max(np.abs(mine["HTH"] - G)) <= 1e-8 * np.max(np.abs(G))
        assert np.max(np.abs(mine["HTz"] - lg[80:92])) <= 1e-8 * np.max(np.abs(lg[80:92]))
        # cond(H^T R^-1 H + P^-1) ~ 1e8 here: the two inversions agree to ~1e-8 relative
        assert np.max(np.abs(mine["sol"] - lg[92:116])) <= 1e-7 * max(np.max(np.abs(lg[92:116])), 1e-3)
        d = oracle.state_boxminus(mine["state"], ref["state"])
        assert np.max(np.abs(d)) <= 1e-9
    # and the default (non-literal) evaluation of the oracle agrees with its literal one
    ref2 = tree.iekf_update(scan, cur, prop, max_iterations=1, imu_en=imu_en, threads=2, literal_gain=False)
    assert np.max(np.abs(ref2["logs"][0][92:116] - lg[92:116])) <= 1e-9
