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
