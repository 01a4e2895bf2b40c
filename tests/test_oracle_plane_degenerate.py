"""esti_plane (reference include/common_lib.h:236-269: A.colPivHouseholderQr().solve(b) on the 5 x 3 neighbour matrix) on
degenerate neighbourhoods, against an INDEPENDENTLY written implementation of Eigen 3.3's published algorithm (numpy, vector
form, different code structure from oracle/orc_plane.hpp and from the kernel's qr_solve_5x3 - the two are the same routine
typed twice and cannot check each other) and, where the rank is unambiguous, against exact rational least squares.

The rank decision: a column counts as zero when its (down-dated) squared norm < threshold_helper * (rows - k) with
threshold_helper = abs2(maxColNorm * eps) / rows  (ColPivHouseholderQR::computeInPlace); solve() uses the first
nonzero_pivots reflectors / columns and sets the other components to zero."""
from fractions import Fraction

import numpy as np


def colpiv_qr_solve_numpy(A, b):
    """Eigen 3.3 ColPivHouseholderQR(A).solve(b), written with numpy vectors (rows x cols = 5 x 3)."""
    qr = np.array(A, np.float64)
    rows, cols = qr.shape
    eps = np.finfo(np.float64).eps
    direct = np.linalg.norm(qr, axis=0)
    upd = direct.copy()
    helper = (upd.max() * eps) ** 2 / rows
    downdate = np.sqrt(eps)
    nz = cols
    perm = list(range(cols))
    taus, c = [], np.array(b, np.float64)
    for k in range(cols):
        big = k + int(np.argmax(upd[k:]))
        if nz == cols and upd[big] ** 2 < helper * (rows - k):
            nz = k
        if big != k:
            qr[:, [k, big]] = qr[:, [big, k]]
            upd[[k, big]] = upd[[big, k]]
            direct[[k, big]] = direct[[big, k]]
            perm[k], perm[big] = perm[big], perm[k]
        x = qr[k:, k].copy()
        tail = float(x[1:] @ x[1:])
        if tail <= np.finfo(np.float64).tiny:
            tau, beta, v = 0.0, x[0], np.r_[1.0, np.zeros(len(x) - 1)]
        else:
            beta = -np.copysign(np.sqrt(x[0] * x[0] + tail), x[0]) if x[0] != 0 else -np.sqrt(tail)
            v = np.r_[1.0, x[1:] / (x[0] - beta)]
            tau = (beta - x[0]) / beta
        taus.append((tau, v))
        qr[k, k] = beta
        qr[k + 1:, k] = v[1:]
        if tau != 0 and k + 1 < cols:
            blk = qr[k:, k + 1:]
            blk -= tau * np.outer(v, v @ blk)
        for j in range(k + 1, cols):
            if upd[j] != 0:
                t = abs(qr[k, j]) / upd[j]
                t = max((1 + t) * (1 - t), 0.0)
                if t * (upd[j] / direct[j]) ** 2 <= downdate:
                    direct[j] = upd[j] = np.linalg.norm(qr[k + 1:, j])
                else:
                    upd[j] *= np.sqrt(t)
    for k in range(nz):
        tau, v = taus[k]
        if tau != 0:
            c[k:] -= tau * v * (v @ c[k:])
    y = np.zeros(cols)
    for i in range(nz - 1, -1, -1):
        y[i] = (c[i] - qr[i, i + 1:nz] @ y[i + 1:nz]) / qr[i, i]
    x = np.zeros(cols)
    for i in range(nz):
        x[perm[i]] = y[i]
    return x, nz


def esti_plane_numpy(pts, threshold=0.1):
    A = np.asarray(pts, np.float32).astype(np.float64).reshape(5, 3)
    n, nz = colpiv_qr_solve_numpy(A, -np.ones(5))
    nn = np.linalg.norm(n)
    with np.errstate(divide="ignore", invalid="ignore"):
        pabcd = np.r_[n / nn, 1.0 / nn]
        ok = bool(np.all(np.abs(A @ pabcd[:3] + pabcd[3]) <= threshold))
    return ok, pabcd, nz


def exact_basic_solution(A, cols):
    """Exact least squares of A[:, cols] x = -1 in rationals (normal equations), zeros elsewhere."""
    F = [[Fraction(float(v)) for v in row] for row in A]
    m = len(cols)
    N = [[sum(F[r][cols[i]] * F[r][cols[j]] for r in range(5)) for j in range(m)] for i in range(m)]
    g = [sum(-F[r][cols[i]] for r in range(5)) for i in range(m)]
    for i in range(m):  # Gauss-Jordan in exact arithmetic
        piv = N[i][i]
        N[i] = [v / piv for v in N[i]]
        g[i] = g[i] / piv
        for r in range(m):
            if r != i:
                f = N[r][i]
                N[r] = [a - f * b for a, b in zip(N[r], N[i])]
                g[r] = g[r] - f * g[i]
    x = [0.0, 0.0, 0.0]
    for i, c in enumerate(cols):
        x[c] = float(g[i])
    return np.array(x)


def test_well_conditioned_neighbourhoods_agree(oracle):
    rng = np.random.default_rng(5)
    for _ in range(300):
        n = rng.normal(0, 1, 3)
        n /= np.linalg.norm(n)
        base = rng.normal(0, 20, 3)
        u = np.cross(n, [1, 0, 0.3]); u /= np.linalg.norm(u)
        v = np.cross(n, u)
        pts = (base + np.outer(rng.uniform(-0.3, 0.3, 5), u) + np.outer(rng.uniform(-0.3, 0.3, 5), v) + np.outer(rng.normal(0, 0.01, 5), n)).astype(np.float32)
        ok_o, p_o = oracle.esti_plane(pts)
        ok_n, p_n, nz = esti_plane_numpy(pts)
        assert nz == 3 and ok_o == ok_n
        assert np.max(np.abs(p_o - p_n)) <= 1e-9 * max(1.0, abs(p_n[3]))
        x = exact_basic_solution(pts.astype(np.float64), [0, 1, 2])
        assert np.max(np.abs(p_n[:3] / p_n[3] - x)) <= 1e-7 * np.max(np.abs(x))  # (cond of a 0.3 m patch 20 m out: ~1e4..1e6)


def test_coplanar_through_origin_axis_planes(oracle):
    """All five points in the plane z = 0 (or x = 0, y = 0): one column of A is exactly zero, the rank is exactly 2, Eigen drops
    the third pivot (nonzero_pivots = 2) and the dropped component of the solution is exactly 0."""
    rng = np.random.default_rng(6)
    for axis in range(3):
        for _ in range(50):
            pts = rng.uniform(-3, 3, (5, 3)).astype(np.float32)
            pts[:, axis] = 0.0
            ok_o, p_o = oracle.esti_plane(pts)
            ok_n, p_n, nz = esti_plane_numpy(pts)
            assert nz == 2
            assert p_o[axis] == 0.0 and p_n[axis] == 0.0
            cols = [c for c in range(3) if c != axis]
            x = exact_basic_solution(pts.astype(np.float64), cols)
            assert np.max(np.abs(p_n[:3] / p_n[3] - x)) <= 1e-10 * max(1.0, np.max(np.abs(x)))
            assert np.max(np.abs(p_o - p_n)) <= 1e-9 * max(1.0, abs(p_n[3]))
            assert ok_o == ok_n


def test_rank_one_and_duplicates(oracle):
    """Five identical points / five points on a line through the origin along an axis: exactly one non-zero column (rank 1)."""
    for axis in range(3):
        pts = np.zeros((5, 3), np.float32)
        pts[:, axis] = [1.0, 2.0, 3.0, 4.0, 5.5]
        ok_o, p_o = oracle.esti_plane(pts)
        ok_n, p_n, nz = esti_plane_numpy(pts)
        assert nz == 1 and ok_o == ok_n
        x = exact_basic_solution(pts.astype(np.float64), [axis])
        assert np.max(np.abs(p_n[:3] / p_n[3] - x)) <= 1e-12
        assert np.max(np.abs(p_o - p_n)) <= 1e-12 * max(1.0, abs(p_n[3]))
    # duplicates off the axes: rank 1 up to rounding - the pivot decision rides on ~eps-sized column remainders; both
    # implementations must take the SAME decision (this is where a wrong threshold_helper shows)
    rng = np.random.default_rng(8)
    agree = 0
    for _ in range(200):
        pts = np.tile(rng.uniform(-30, 30, 3).astype(np.float32), (5, 1))
        ok_o, p_o = oracle.esti_plane(pts)
        ok_n, p_n, nz = esti_plane_numpy(pts)
        same = np.allclose(p_o, p_n, rtol=1e-6, atol=1e-9, equal_nan=True)
        agree += int(same and ok_o == ok_n)
    assert agree >= 150  # (the remainders are rounding noise: the two codes round differently in ~10 % of the cases)


def test_four_coplanar_plus_one(oracle):
    rng = np.random.default_rng(9)
    for _ in range(100):
        pts = np.c_[rng.uniform(-1, 1, (5, 2)), np.full(5, 2.0)].astype(np.float32)
        pts[4, 2] += np.float32(rng.choice([0.05, 0.3, -0.2]))
        ok_o, p_o = oracle.esti_plane(pts)
        ok_n, p_n, nz = esti_plane_numpy(pts)
        assert nz == 3 and ok_o == ok_n
        assert np.max(np.abs(p_o - p_n)) <= 1e-10 * max(1.0, abs(p_n[3]))
