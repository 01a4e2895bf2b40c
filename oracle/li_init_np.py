"""ORACLE — TEST INFRASTRUCTURE ONLY.

numpy restatement of the reference's batch LiDAR-IMU initialisation (include/LI_init/LI_init.{h,cpp}):
the signal-conditioning chain with every quirk the reference has, the three cost
functors with analytic Jacobians, and a Levenberg-Marquardt loop that follows Ceres 2.0's documented
trust-region algorithm (Ceres is third-party, pinned at 2.0.0 by docker/Dockerfile:16-21 and NOT vendored).

  CalibSeq ..................... CalibState deques                        LI_init.h:31-89 (operator= quirk A7)
  imu_time_compensate .......... LI_Init::IMU_time_compensate             LI_init.cpp:195-221
  cut_sequence_tail ............ LI_Init::cut_sequence_tail               LI_init.cpp:223-238
  butter_filt / zero_phase_filt  LI_Init::Butter_filt / zero_phase_filt   LI_init.cpp:260-315, coefficients LI_init.h:218-224
  normalize_acc ................ LI_Init::normalize_acc                   LI_init.cpp:494-504
  xcorr_temporal_init .......... LI_Init::xcorr_temporal_init             LI_init.cpp:160-193
  central_diff ................. LI_Init::central_diff                    LI_init.cpp:127-158
  acc_interpolate .............. LI_Init::acc_interpolate                 LI_init.cpp:240-258
  residuals_stage{1,2,3} ....... Angular_Vel_Cost_only_Rot / Angular_Vel_Cost / Linear_acc_Cost   LI_init.h:91-205
  solve_stage{1,2,3} ........... solve_Rotation_only / solve_Rot_bias_gyro / solve_trans_biasacc_grav  LI_init.cpp:317-492
  li_initialization ............ LI_Init::LI_Initialization (from the IMU_time_compensate(0, true) step on)  LI_init.cpp:593-625

Pinned against the reference's own committed run (Log/*.txt, result/Initialization_result.txt) by
tests/test_oracle_li_init.py through the fixtures in tests/golden/li_init/.
"""
from __future__ import annotations

import numpy as np

G_m_s2 = 9.81
STD_GRAV = np.array([0.0, 0.0, -G_m_s2])  # LI_init.h:27

# LI_init.h:218-224 — note Coeff_b[4] = 0.0011, not the symmetric 0.001143 (quirk A8)
COEFF_B = np.array([0.000076, 0.000457, 0.001143, 0.001524, 0.0011, 0.000457, 0.000076])
COEFF_A = np.array([1.0000, -4.182389, 7.491611, -7.313596, 4.089349, -1.238525, 0.158428])


class CalibSeq:
    """A deque<CalibState> as arrays. Fields: t, ang_vel, linear_vel, ang_acc, linear_acc, rot_end."""

    FIELDS4 = ("ang_vel", "linear_vel", "ang_acc", "linear_acc")

    def __init__(self, n=0):
        self.t = np.zeros(n)
        self.ang_vel = np.zeros((n, 3))
        self.linear_vel = np.zeros((n, 3))
        self.ang_acc = np.zeros((n, 3))
        self.linear_acc = np.zeros((n, 3))
        self.rot_end = np.tile(np.eye(3), (n, 1, 1))

    def __len__(self):
        return len(self.t)

    def copy(self):
        return self.slice(slice(None))

    def slice(self, s):
        o = CalibSeq(0)
        o.t = self.t[s].copy()
        for f in self.FIELDS4:
            setattr(o, f, getattr(self, f)[s].copy())
        o.rot_end = self.rot_end[s].copy()
        return o

    def to_records(self):
        """n x 22 doubles in the lii_calib_state layout: rot_end[9], ang_vel, linear_vel, ang_acc, linear_acc, timestamp."""
        n = len(self)
        out = np.zeros((n, 22))
        out[:, 0:9] = self.rot_end.reshape(n, 9)
        out[:, 9:12] = self.ang_vel
        out[:, 12:15] = self.linear_vel
        out[:, 15:18] = self.ang_acc
        out[:, 18:21] = self.linear_acc
        out[:, 21] = self.t
        return out


# ---------------------------------------------------------------------------------------------- conditioning chain
def _align(imu: CalibSeq, lidar: CalibSeq):
    """The pop-front / pop-back alignment shared by IMU_time_compensate and cut_sequence_tail (:213-221, :228-237)."""
    li, ii = 0, 0
    while lidar.t[li] < imu.t[ii]:
        li += 1
    while lidar.t[li] > imu.t[ii + 1]:
        ii += 1
    imu, lidar = imu.slice(slice(ii, None)), lidar.slice(slice(li, None))
    n = min(len(imu), len(lidar))
    return imu.slice(slice(0, n)), lidar.slice(slice(0, n))


def imu_time_compensate(imu: CalibSeq, lidar: CalibSeq, lag_time: float, is_discard: bool):
    if is_discard:  # the first 10 pairs are dropped on the first call (:196-203)
        imu, lidar = imu.slice(slice(10, None)), lidar.slice(slice(10, None))
    else:
        imu, lidar = imu.copy(), lidar.copy()
    imu.t[:-1] = imu.t[:-1] - lag_time  # the LAST stamp is left unshifted (`!= end() - 1`, :206-208, quirk A9)
    return _align(imu, lidar)


def cut_sequence_tail(imu: CalibSeq, lidar: CalibSeq):
    return _align(imu.slice(slice(0, len(imu) - 20)), lidar.slice(slice(0, len(lidar) - 20)))


def butter_filt(sig: CalibSeq) -> CalibSeq:
    """One pass of the 6th-order Butterworth over the four 3-vector fields.  CalibState::operator= copies only those
    fields (quirk A7), so timeStamp / rot_end of the padded input survive."""
    ext = 10 * (len(COEFF_B) - 1)  # 60
    n = len(sig)
    out = CalibSeq(0)
    # reflection padding: front = sig[ext], sig[ext-1], ..., sig[1] reversed into place; back mirrored likewise (:266-286)
    front = np.arange(ext, 0, -1)
    back = np.arange(n - 2, n - 2 - ext, -1)
    idx = np.r_[front, np.arange(n), back]
    t_ext = sig.t[idx]
    rot_ext = sig.rot_end[idx]
    res = {}
    for f in CalibSeq.FIELDS4:
        x = getattr(sig, f)[idx]
        y = x.copy()
        nb = len(COEFF_B)
        for i in range(nb, len(x) - ext):  # starts at Coeff_size (7), stops extend_num before the end (:289)
            acc = np.zeros(3)
            for j in range(nb):
                acc += x[i - j] * COEFF_B[j]
            for jj in range(1, nb):
                acc -= y[i - jj] * COEFF_A[jj]
            y[i] = acc
        res[f] = y[ext:len(x) - ext]
    out.t = t_ext[ext:len(idx) - ext].copy()
    out.rot_end = rot_ext[ext:len(idx) - ext].copy()
    for f in CalibSeq.FIELDS4:
        setattr(out, f, res[f].copy())
    return out


def _reverse(sig: CalibSeq) -> CalibSeq:
    return sig.slice(slice(None, None, -1))


def zero_phase_filt(sig: CalibSeq) -> CalibSeq:
    return _reverse(butter_filt(_reverse(butter_filt(sig))))


def normalize_acc(sig: CalibSeq) -> CalibSeq:
    sig = sig.copy()
    mean_acc = np.zeros(3)
    for i in range(1, 10):  # samples 1..9, running mean (:497-499)
        mean_acc += (sig.linear_acc[i] - mean_acc) / i
    sig.linear_acc = sig.linear_acc / np.linalg.norm(mean_acc) * G_m_s2
    return sig


def xcorr_temporal_init(imu: CalibSeq, lidar: CalibSeq, odom_freq: float):
    """Zero-centred cross-correlation of |omega|; returns (time_lag_1, lag_IMU_wtr_Lidar)."""
    a = np.linalg.norm(imu.ang_vel, axis=1)
    b = np.linalg.norm(lidar.ang_vel, axis=1)
    n = len(a)
    ma, mb = 0.0, 0.0
    for i in range(n):  # running means exactly as :164-167
        ma += (a[i] - ma) / (i + 1)
        mb += (b[i] - mb) / (i + 1)
    a0, b0 = a - ma, b - mb
    # corr(lag) = sum_i a0[i] b0[i + lag]; first maximum wins (`corr > max_corr`)
    full = np.correlate(b0, a0, mode="full")  # index k <-> lag = k - (n - 1)
    lag = int(np.argmax(full)) - (n - 1)
    lag_imu_wtr_lidar = -lag
    return lag_imu_wtr_lidar / odom_freq, lag_imu_wtr_lidar


def central_diff(imu: CalibSeq, lidar: CalibSeq):
    """In place on copies: ang_acc (both) and lidar linear_acc for indices 1 .. n-3 (:127-158)."""
    imu, lidar = imu.copy(), lidar.copy()
    n = len(imu)
    for s in (imu, lidar):
        dt = (s.t[2:n - 1] - s.t[0:n - 3])[:, None]
        s.ang_acc[1:n - 2] = (s.ang_vel[2:n - 1] - s.ang_vel[0:n - 3]) / dt
    dt = (lidar.t[2:n - 1] - lidar.t[0:n - 3])[:, None]
    lidar.linear_acc[1:n - 2] = (lidar.linear_vel[2:n - 1] - lidar.linear_vel[0:n - 3]) / dt
    return imu, lidar


def acc_interpolate(imu: CalibSeq, lidar: CalibSeq) -> CalibSeq:
    """Sequential, in place like the reference (uses already-updated i-1 for deltaT <= 0) (:240-258)."""
    imu = imu.copy()
    for i in range(1, len(lidar) - 1):
        d = lidar.t[i] - imu.t[i]
        if d > 0:
            D = imu.t[i + 1] - imu.t[i]
            s = d / D
            imu.linear_acc[i] = s * imu.linear_acc[i + 1] + (1 - s) * imu.linear_acc[i]
        else:
            D = imu.t[i] - imu.t[i - 1]
            s = -d / D
            imu.linear_acc[i] = s * imu.linear_acc[i - 1] + (1 - s) * imu.linear_acc[i]
        imu.t[i] += d
    return imu


def downsample_interpolate_imu(imu_all: CalibSeq, lidar: CalibSeq, move_start_time: float):
    """LI_Init::downsample_interpolate_IMU (:82-125): drop everything older than move_start - 3 s, 5-tap running-mean filter
    of the accelerations, linear interpolation of omega / acc at the LiDAR stamps.  The reference's filter reads one element
    past the end of its copy in its last iteration (UB, quirk A10); that read is the last raw sample here."""
    ka = int(np.argmax(imu_all.t >= move_start_time - 3.0)) if np.any(imu_all.t >= move_start_time - 3.0) else len(imu_all)
    kl = int(np.argmax(lidar.t >= move_start_time - 3.0)) if np.any(lidar.t >= move_start_time - 3.0) else len(lidar)
    a, l = imu_all.slice(slice(ka, None)), lidar.slice(slice(kl, None))
    origin = a.linear_acc.copy()
    for i in range(2, len(a) - 2):
        acc = np.zeros(3)
        for d in range(-2, 3):
            acc += (origin[i + d] - acc) / (d + 3)
        a.linear_acc[i] = acc
    out_i, keep = [], []
    for i in range(len(l)):
        j = np.searchsorted(a.t, l.t[i], side="right")  # first j with a.t[j] > t ; need a.t[j-1] <= t
        if j < 1 or j >= len(a):
            continue
        s_ = (a.t[j] - l.t[i]) / (a.t[j] - a.t[j - 1])
        out_i.append((s_ * a.ang_vel[j - 1] + (1 - s_) * a.ang_vel[j], s_ * a.linear_acc[j - 1] + (1 - s_) * a.linear_acc[j], l.t[i]))
        keep.append(i)
    imu = CalibSeq(len(out_i))
    for k, (w, acc, t) in enumerate(out_i):
        imu.ang_vel[k], imu.linear_acc[k], imu.t[k] = w, acc, t
    return imu, l.slice(np.array(keep, dtype=int))


# ---------------------------------------------------------------------------------------------- residuals + Jacobians
def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


def quat_to_rot(q):  # Eigen::Quaternion::toRotationMatrix, q = (w, x, y, z)
    w, x, y, z = q
    tx, ty, tz = 2 * x, 2 * y, 2 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    return np.array([[1 - (tyy + tzz), txy - twz, txz + twy],
                     [txy + twz, 1 - (txx + tzz), tyz - twx],
                     [txz - twy, tyz + twx, 1 - (txx + tyy)]])


def rot_to_quat(m):  # Eigen::Quaternion(Matrix3)
    t = np.trace(m)
    if t > 0:
        t = np.sqrt(t + 1.0)
        w = 0.5 * t
        t = 0.5 / t
        return np.array([w, (m[2, 1] - m[1, 2]) * t, (m[0, 2] - m[2, 0]) * t, (m[1, 0] - m[0, 1]) * t])
    i = 0
    if m[1, 1] > m[0, 0]:
        i = 1
    if m[2, 2] > m[i, i]:
        i = 2
    j, k = (i + 1) % 3, (i + 2) % 3
    t = np.sqrt(m[i, i] - m[j, j] - m[k, k] + 1.0)
    v = np.zeros(3)
    v[i] = 0.5 * t
    t = 0.5 / t
    w = (m[k, j] - m[j, k]) * t
    v[j] = (m[j, i] + m[i, j]) * t
    v[k] = (m[k, i] + m[i, k]) * t
    return np.array([w, v[0], v[1], v[2]])


def quat_plus(q, d):  # ceres::QuaternionParameterization::Plus
    n = np.linalg.norm(d)
    if n <= 0.0:
        return q.copy()
    s = np.sin(n) / n
    a, b, c, e = np.cos(n), s * d[0], s * d[1], s * d[2]
    w, x, y, z = q
    return np.array([a * w - b * x - c * y - e * z, a * x + b * w + c * z - e * y, a * y - b * z + c * w + e * x,
                     a * z + b * y - c * x + e * w])


def normal_equations(stage, R, v, imu: CalibSeq, lidar: CalibSeq, R_LI=None):
    """J^T J, J^T r, cost = 0.5 sum r^2 in the tangent convention R <- Exp(delta) R (3 / 7 / 9 dof)."""
    n = len(imu)
    if stage in (1, 2):
        Rw = lidar.ang_vel @ R.T
        r = Rw - imu.ang_vel
        dof = 3 if stage == 1 else 7
        J = np.zeros((n, 3, dof))
        J[:, 0, 1], J[:, 0, 2] = Rw[:, 2], -Rw[:, 1]
        J[:, 1, 0], J[:, 1, 2] = -Rw[:, 2], Rw[:, 0]
        J[:, 2, 0], J[:, 2, 1] = Rw[:, 1], -Rw[:, 0]
        if stage == 2:
            bg, td = v[0:3], v[3]
            dT = lidar.t - imu.t
            r = r - (dT + td)[:, None] * imu.ang_acc + bg
            J[:, 0, 3] = J[:, 1, 4] = J[:, 2, 5] = 1.0
            J[:, :, 6] = -imu.ang_acc
    else:
        ba, Til = v[0:3], v[3:6]
        RLL0 = lidar.rot_end
        aI_L = imu.linear_acc @ R_LI  # R_LI^T a_I
        t1 = np.einsum("nij,nj->ni", RLL0, aI_L)
        Rg = R @ STD_GRAV
        w, al = lidar.ang_vel, lidar.ang_acc
        W = np.zeros((n, 3, 3))
        A = np.zeros((n, 3, 3))
        for M_, x in ((W, w), (A, al)):
            M_[:, 0, 1], M_[:, 0, 2] = -x[:, 2], x[:, 1]
            M_[:, 1, 0], M_[:, 1, 2] = x[:, 2], -x[:, 0]
            M_[:, 2, 0], M_[:, 2, 1] = -x[:, 1], x[:, 0]
        M = W @ W + A
        RM = RLL0 @ M
        r = t1 - RLL0 @ ba + Rg - lidar.linear_acc - RM @ Til
        J = np.zeros((n, 3, 9))
        J[:, :, 0:3] = -skew(Rg)
        J[:, :, 3:6] = -RLL0
        J[:, :, 6:9] = -RM
    JtJ = np.einsum("nak,nal->kl", J, J)
    Jtr = np.einsum("nak,na->k", J, r)
    return JtJ, Jtr, 0.5 * float(np.sum(r * r))


class _Problem:
    def __init__(self, stage, q, v, imu, lidar, R_LI=None, lo=None, hi=None):
        self.stage, self.q, self.v = stage, np.array(q, float), np.array(v, float)
        self.imu, self.lidar, self.R_LI = imu, lidar, R_LI
        self.lo, self.hi = lo, hi
        self.dof = 3 + len(self.v)

    def eval(self, need_jac=True):
        JtJ, Jtr, cost = normal_equations(self.stage, quat_to_rot(self.q), self.v, self.imu, self.lidar, self.R_LI)
        if need_jac:
            s = np.ones(self.dof)
            s[:3] = 2.0  # Ceres' quaternion tangent rotates by 2|delta|
            JtJ = JtJ * s[:, None] * s[None, :]
            Jtr = Jtr * s
        return JtJ, Jtr, cost

    def plus(self, d):
        c = _Problem(self.stage, quat_plus(self.q, d[:3]), self.v + d[3:], self.imu, self.lidar, self.R_LI, self.lo, self.hi)
        if self.lo is not None:
            m = np.isfinite(self.lo)
            c.v[m] = np.minimum(np.maximum(c.v[m], self.lo[m]), self.hi[m])  # ParameterBlock::Plus projects onto the box
        return c

    def ambient(self):
        return np.r_[self.q, self.v]


def ceres_like_lm(p: _Problem, max_iterations=50, verbose=False):
    """Ceres 2.0 TrustRegionMinimizer + LevenbergMarquardtStrategy with default Solver::Options (see module doc of
    lidar_imu_init_amd/csrc/lii_calib.cpp, which implements the same loop around the HIP evaluator)."""
    ftol, gtol, ptol = 1e-6, 1e-10, 1e-8
    radius, decrease = 1e4, 2.0
    p = p.plus(np.zeros(p.dof))
    JtJ, g, cost = p.eval()
    scale = 1.0 / (1.0 + np.sqrt(np.diag(JtJ)))

    def gmax(pp, gg):
        return np.max(np.abs(pp.ambient() - pp.plus(-gg).ambient()))

    if gmax(p, g) <= gtol:
        return p, cost, 0
    x_norm = np.linalg.norm(p.ambient())
    it, invalid, reuse = 0, 0, False
    diag = None
    while it < max_iterations and radius >= 1e-32:
        it += 1
        Js = JtJ * scale[:, None] * scale[None, :]
        gs = g * scale
        if not reuse:
            diag = np.clip(np.diag(Js), 1e-6, 1e32)
        reuse = True
        try:
            L = np.linalg.cholesky(Js + np.diag(diag / radius))
            step = -np.linalg.solve(L.T, np.linalg.solve(L, gs))
            model = -step @ gs - 0.5 * step @ Js @ step
            ok = model > 0
        except np.linalg.LinAlgError:
            ok = False
        if not ok:
            invalid += 1
            if invalid >= 5:
                break
            radius /= decrease
            decrease *= 2
            continue
        invalid = 0
        cand = p.plus(step * scale)
        _, _, ccost = cand.eval(False)
        if np.linalg.norm(p.ambient() - cand.ambient()) <= ptol * (x_norm + ptol):
            break
        change = cost - ccost
        if abs(change) <= ftol * cost:
            break
        rho = change / model
        if verbose:
            print(f"  it {it} cost {cost:.9e} -> {ccost:.9e} rho {rho:.3f} radius {radius:.3e}")
        if rho > 1e-3:
            p = cand
            x_norm = np.linalg.norm(p.ambient())
            JtJ, g, cost = p.eval()
            radius = min(1e16, radius / max(1.0 / 3.0, 1.0 - (2.0 * rho - 1.0) ** 3))
            decrease = 2.0
            reuse = False
            if gmax(p, g) <= gtol:
                break
        else:
            radius /= decrease
            decrease *= 2
    return p, cost, it


def solve_stage1(imu, lidar):
    p, cost, it = ceres_like_lm(_Problem(1, [1, 0, 0, 0], [], imu, lidar))
    return dict(R_LI=quat_to_rot(p.q), cost=cost, iterations=it)


def solve_stage2(imu, lidar, R_init):
    p, cost, it = ceres_like_lm(_Problem(2, rot_to_quat(R_init), [0, 0, 0, 0], imu, lidar))
    return dict(R_LI=quat_to_rot(p.q), gyro_bias=p.v[0:3].copy(), time_lag_2=float(p.v[3]), cost=cost, iterations=it)


def solve_stage3(imu, lidar, R_LI):
    lo = np.r_[-0.01, -0.01, -0.01, -np.inf, -np.inf, -np.inf]
    hi = np.r_[0.01, 0.01, 0.01, np.inf, np.inf, np.inf]
    p, cost, it = ceres_like_lm(_Problem(3, [1, 0, 0, 0], np.zeros(6), imu, lidar, R_LI, lo, hi))
    R_GL0 = quat_to_rot(p.q)
    return dict(R_GL0=R_GL0, grav_L0=R_GL0 @ STD_GRAV, acc_bias=R_LI @ p.v[0:3], T_LI=-R_LI @ p.v[3:6],
                bias_aL=p.v[0:3].copy(), T_IL=p.v[3:6].copy(), cost=cost, iterations=it)


# ---------------------------------------------------------------------------------------------- the driver
def li_initialization(imu0: CalibSeq, lidar0: CalibSeq, orig_odom_freq: int, cut_frame_num: int, solve=True):
    """LI_Init::LI_Initialization from `IMU_time_compensate(0.0, true)` on (LI_init.cpp:593-625).  imu0 / lidar0 are the
    sequences after downsample_interpolate_IMU (what fout_before_filter dumps, plus the last element it omits)."""
    out = {}
    imu, lidar = imu_time_compensate(imu0, lidar0, 0.0, True)
    imu_f = normalize_acc(zero_phase_filt(imu))
    lidar_f = zero_phase_filt(lidar)
    imu, lidar = imu_f.slice(slice(0, len(imu_f) - 1)), lidar_f.slice(slice(0, len(lidar_f) - 1))  # set_*_state drop the last
    imu, lidar = cut_sequence_tail(imu, lidar)
    lag1, lag_frames = xcorr_temporal_init(imu, lidar, orig_odom_freq * cut_frame_num)
    out["time_lag_1"], out["lag_frames"] = lag1, lag_frames
    imu, lidar = imu_time_compensate(imu, lidar, lag1, False)
    imu, lidar = central_diff(imu, lidar)
    out["imu_meas"], out["lidar_meas"] = imu.copy(), lidar.copy()
    imu2, lidar2 = zero_phase_filt(imu), zero_phase_filt(lidar)
    imu.ang_acc = imu2.ang_acc.copy()            # set_states_2nd_filter (:35-41)
    lidar.ang_acc = lidar2.ang_acc.copy()
    lidar.linear_acc = lidar2.linear_acc.copy()
    out["imu_stage12"], out["lidar_stage12"] = imu.copy(), lidar.copy()
    if not solve:
        return out
    s1 = solve_stage1(imu, lidar)
    s2 = solve_stage2(imu, lidar, s1["R_LI"])
    out["stage1"], out["stage2"] = s1, s2
    imu, lidar = imu_time_compensate(imu, lidar, s2["time_lag_2"], False)  # second temporal compensation (:395)
    out["lidar_after_rot"] = (lidar.ang_vel @ s2["R_LI"].T + s2["gyro_bias"], lidar.t.copy())
    imu = acc_interpolate(imu, lidar)
    out["imu_stage3"], out["lidar_stage3"] = imu.copy(), lidar.copy()
    out["stage3"] = solve_stage3(imu, lidar, s2["R_LI"])
    out["time_delay"] = lag1 + s2["time_lag_2"]
    return out
