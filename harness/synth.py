"""Seeded synthetic worlds, LiDAR scans and local maps for tests and bench.py.

No datasets ship with the reference (its rosbags live on a Google Drive, README.md:132-138) and there is no
network here, so every workload is generated: a closed piecewise-planar hall (floor, ceiling, walls, box
obstacles), spinning-LiDAR ray casting with Gaussian range noise (which also breaks k-NN ties) and a
per-point time offset in milliseconds derived from the azimuth (the reference's `curvature` field).
Pure numpy; deterministic for a given seed.
"""
from __future__ import annotations

import numpy as np

SEED = 20220613


def rot_zyx(roll, pitch, yaw):
    cr, sr, cp, sp, cy, sy = np.cos(roll), np.sin(roll), np.cos(pitch), np.sin(pitch), np.cos(yaw), np.sin(yaw)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


class Hall:
    """Axis-aligned hall [lo, hi] with axis-aligned box obstacles standing inside it."""

    def __init__(self, size=(40.0, 30.0, 8.0), n_boxes=12, seed=SEED):
        rng = np.random.default_rng(seed)
        self.lo = np.array([-size[0] / 2, -size[1] / 2, -1.5])
        self.hi = np.array([size[0] / 2, size[1] / 2, size[2] - 1.5])
        boxes = []
        for _ in range(n_boxes):
            c = rng.uniform(self.lo[:2] + 2.0, self.hi[:2] - 2.0)
            if np.linalg.norm(c) < 4.0:  # keep the start area free
                c = c / max(np.linalg.norm(c), 1e-3) * 5.0
            w = rng.uniform(0.6, 3.0, 2)
            hgt = rng.uniform(0.8, size[2] * 0.7)
            boxes.append((np.array([c[0] - w[0] / 2, c[1] - w[1] / 2, self.lo[2]]),
                          np.array([c[0] + w[0] / 2, c[1] + w[1] / 2, self.lo[2] + hgt])))
        self.boxes = boxes

    # ---------------------------------------------------------------- ray casting
    def raycast(self, origin, dirs):
        """Distance along each unit direction from `origin` (inside the hall) to the first surface."""
        o = np.asarray(origin, np.float64)
        d = np.asarray(dirs, np.float64)
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = 1.0 / d
            # exit distance from the enclosing hall
            t1 = (self.lo - o) * inv
            t2 = (self.hi - o) * inv
            t_exit = np.nanmin(np.where(d != 0, np.maximum(t1, t2), np.inf), axis=1)
            best = t_exit
            for lo, hi in self.boxes:
                a = (lo - o) * inv
                b = (hi - o) * inv
                tn = np.nanmax(np.where(d != 0, np.minimum(a, b), -np.inf), axis=1)
                tf = np.nanmin(np.where(d != 0, np.maximum(a, b), np.inf), axis=1)
                hit = (tn <= tf) & (tn > 1e-6)
                best = np.where(hit & (tn < best), tn, best)
        return best

    # ---------------------------------------------------------------- surface sampling (local map)
    def surface_points(self, spacing, noise=0.01, seed=SEED, max_points=None):
        """Points on every surface on a jittered lattice of the given spacing: the shape of a converged
        ikd-Tree map (about one point per `filter_size_map` voxel)."""
        rng = np.random.default_rng(seed + 1)
        pts = []

        def face(lo, hi, axis, value):
            ax = [a for a in range(3) if a != axis]
            u = np.arange(lo[ax[0]] + spacing / 2, hi[ax[0]], spacing)
            v = np.arange(lo[ax[1]] + spacing / 2, hi[ax[1]], spacing)
            if len(u) == 0 or len(v) == 0:
                return
            U, V = np.meshgrid(u, v, indexing="ij")
            P = np.zeros((U.size, 3))
            P[:, ax[0]] = U.ravel() + rng.uniform(-0.3, 0.3, U.size) * spacing
            P[:, ax[1]] = V.ravel() + rng.uniform(-0.3, 0.3, U.size) * spacing
            P[:, axis] = value + rng.normal(0, noise, U.size)
            pts.append(P)

        for axis in range(3):
            face(self.lo, self.hi, axis, self.lo[axis])
            face(self.lo, self.hi, axis, self.hi[axis])
        for lo, hi in self.boxes:
            for axis in range(3):
                if axis == 2:
                    face(lo, hi, axis, hi[axis])  # top only
                else:
                    face(lo, hi, axis, lo[axis])
                    face(lo, hi, axis, hi[axis])
        P = np.concatenate(pts).astype(np.float32)
        if max_points is not None and len(P) > max_points:
            P = P[rng.choice(len(P), max_points, replace=False)]
        return P


def spinning_lidar(n_rings, n_cols, fov_down_deg, fov_up_deg, sweep_ms=100.0):
    """Unit directions (rings x cols) and per-point time offsets [ms] of one sweep of a spinning LiDAR."""
    el = np.deg2rad(np.linspace(fov_down_deg, fov_up_deg, n_rings))
    az = np.linspace(0, 2 * np.pi, n_cols, endpoint=False)
    EL, AZ = np.meshgrid(el, az, indexing="ij")
    dirs = np.stack([np.cos(EL) * np.cos(AZ), np.cos(EL) * np.sin(AZ), np.sin(EL)], -1).reshape(-1, 3)
    t_ms = np.broadcast_to(az / (2 * np.pi) * sweep_ms, EL.shape).reshape(-1).astype(np.float32)
    return dirs, t_ms


SENSORS = {
    # name: (rings, columns, fov_down, fov_up)   -> points per sweep
    "vlp16": (16, 1875, -15.0, 15.0),      # ~30 k  (BASELINE.json configs[1])
    "os1_128": (128, 1024, -22.5, 22.5),   # 131 072 (configs[2])
    "stream100k": (100, 1000, -25.0, 25.0),  # 100 000 (north-star stream)
    "dense500k": (128, 3906, -25.0, 15.0),   # ~500 k (configs[4])
    "tiny": (16, 128, -15.0, 15.0),        # 2 048 — unit tests
    "mid16k": (32, 512, -25.0, 25.0),      # 16 384 — end-to-end plumbing test
}


def make_scan(hall: Hall, sensor: str, R_wb, p_wb, noise=0.02, seed=SEED, max_range=100.0, blind=0.5):
    """One undistorted sweep taken from body pose (R_wb, p_wb): float32 (n,4) = body-frame xyz + t_ms."""
    rings, cols, fd, fu = SENSORS[sensor]
    rng = np.random.default_rng(seed + 7)
    dirs_b, t_ms = spinning_lidar(rings, cols, fd, fu)
    dirs_w = dirs_b @ np.asarray(R_wb).T
    rngs = hall.raycast(p_wb, dirs_w)
    rngs = rngs + rng.normal(0, noise, len(rngs))
    ok = np.isfinite(rngs) & (rngs > blind) & (rngs < max_range)
    pts = dirs_b[ok] * rngs[ok, None]
    return np.concatenate([pts, t_ms[ok, None]], 1).astype(np.float32)


def bench_world(n_map_points=1_000_000, spacing=0.15, seed=SEED):
    """A hall large enough that its surfaces hold `n_map_points` map points at the given spacing."""
    area = n_map_points * spacing * spacing  # m^2 of surface needed
    # floor + ceiling dominate: 2*L*W + 2*(L+W)*H  with H = 10, L = 1.6 W
    H = 10.0
    W = (-2 * 2.6 * H + np.sqrt((2 * 2.6 * H) ** 2 + 4 * 3.2 * area)) / (2 * 3.2)
    L = 1.6 * W
    hall = Hall(size=(L, W, H), n_boxes=40, seed=seed)
    pts = hall.surface_points(spacing, noise=0.01, seed=seed)
    rng = np.random.default_rng(seed + 3)
    if len(pts) > n_map_points:
        pts = pts[rng.choice(len(pts), n_map_points, replace=False)]
    return hall, pts


# ------------------------------------------------------------------------------------------------ moving platform
def rot_zyx_batch(roll, pitch, yaw):
    cr, sr, cp, sp, cy, sy = np.cos(roll), np.sin(roll), np.cos(pitch), np.sin(pitch), np.cos(yaw), np.sin(yaw)
    R = np.empty(roll.shape + (3, 3))
    R[..., 0, 0] = cy * cp; R[..., 0, 1] = cy * sp * sr - sy * cr; R[..., 0, 2] = cy * sp * cr + sy * sr
    R[..., 1, 0] = sy * cp; R[..., 1, 1] = sy * sp * sr + cy * cr; R[..., 1, 2] = sy * sp * cr - cy * sr
    R[..., 2, 0] = -sp;     R[..., 2, 1] = cp * sr;                R[..., 2, 2] = cp * cr
    return R


class Trajectory:
    """Smooth 6-DoF motion of the LiDAR frame with >= 0.5 rad/s excitation about all three axes after a soft start
    (LI_Init::data_sufficiency_assess needs rotation about every axis, LI_init.cpp:531-556; README.md:97 asks for a standstill)."""

    def __init__(self, ramp_s=2.0, amp=(0.32, 0.30, 0.45), freq=(0.37, 0.29, 0.23), pamp=(1.2, 0.9, 0.25), pfreq=(0.21, 0.17, 0.31)):
        self.ramp, self.amp, self.freq, self.pamp, self.pfreq = ramp_s, np.array(amp), np.array(freq), np.array(pamp), np.array(pfreq)

    def _gain(self, t):
        x = np.clip(np.asarray(t, float) / self.ramp, 0.0, 1.0)
        return x * x * x * (x * (6 * x - 15) + 10)  # smootherstep: zero velocity and acceleration at both ends

    def euler(self, t):
        t = np.asarray(t, float)
        g = self._gain(t)
        return [g * self.amp[k] * np.sin(2 * np.pi * self.freq[k] * t + 0.4 * k) for k in range(3)]

    def R(self, t):
        r, p, y = self.euler(t)
        return rot_zyx_batch(np.asarray(r), np.asarray(p), np.asarray(y))

    def p(self, t):
        t = np.asarray(t, float)
        g = self._gain(t)
        return np.stack([g * self.pamp[k] * np.sin(2 * np.pi * self.pfreq[k] * t + 0.9 * k) for k in range(3)], -1)

    def omega_body(self, t, h=1e-4):
        """vee(R^T dR/dt) by a symmetric difference of the rotation."""
        Rm, Rp = self.R(np.asarray(t) - h), self.R(np.asarray(t) + h)
        D = np.einsum("...ji,...jk->...ik", Rm, Rp)  # R(t-h)^T R(t+h) ~ Exp(2 h w)
        w = np.stack([D[..., 2, 1] - D[..., 1, 2], D[..., 0, 2] - D[..., 2, 0], D[..., 1, 0] - D[..., 0, 1]], -1) / 2.0
        return w / (2 * h)

    def vel(self, t, h=1e-4):
        return (self.p(np.asarray(t) + h) - self.p(np.asarray(t) - h)) / (2 * h)


def make_distorted_scan(hall: Hall, sensor: str, traj: Trajectory, t_beg: float, sweep_s: float, noise=0.01, seed=SEED,
                        blind=0.5, max_range=100.0):
    """One sub-frame taken WHILE the platform moves: ray j is cast at time t_beg + frac_j * sweep_s from the pose the
    platform has at that instant; the returned float4 cloud holds the raw (skewed) body-frame points and their time
    offsets in ms — what the de-skew kernels have to undo."""
    rings, cols, fd, fu = SENSORS[sensor]
    rng = np.random.default_rng(seed)
    dirs_b, t_ms = spinning_lidar(rings, cols, fd, fu, sweep_ms=1000.0 * sweep_s)
    tj = t_beg + t_ms.astype(np.float64) / 1000.0
    R = traj.R(tj)
    o = traj.p(tj)
    dirs_w = np.einsum("nij,nj->ni", R, dirs_b)
    rngs = hall.raycast(o, dirs_w) + rng.normal(0, noise, len(dirs_b))
    ok = np.isfinite(rngs) & (rngs > blind) & (rngs < max_range)
    pts = dirs_b[ok] * rngs[ok, None]
    return np.concatenate([pts, t_ms[ok, None]], 1).astype(np.float32)


def simulate_imu(traj: Trajectory, t0, t1, rate_hz, R_LI, T_LI, b_g, b_a, t_offset, noise_g=1e-3, noise_a=1e-2, seed=SEED):
    """IMU rigidly mounted on the LiDAR with p_I = R_LI p_L + T_LI.  Returns stamps (true time + t_offset), gyro, accel."""
    rng = np.random.default_rng(seed + 11)
    t = np.arange(t0, t1, 1.0 / rate_hz)
    R_WL = traj.R(t)
    w_L = traj.omega_body(t)
    gyro = w_L @ np.asarray(R_LI).T + b_g + rng.normal(0, noise_g, (len(t), 3))
    T_IL = -np.asarray(R_LI).T @ np.asarray(T_LI)  # IMU origin in the LiDAR frame
    h = 2e-3

    def p_I(tt):
        return traj.p(tt) + np.einsum("nij,j->ni", traj.R(tt), T_IL)

    acc_w = (p_I(t + h) - 2 * p_I(t) + p_I(t - h)) / (h * h)
    g_w = np.array([0.0, 0.0, -9.81])
    f_L = np.einsum("nji,nj->ni", R_WL, acc_w - g_w)  # specific force in the LiDAR frame
    accel = f_L @ np.asarray(R_LI).T + b_a + rng.normal(0, noise_a, (len(t), 3))
    return t + t_offset, gyro, accel
