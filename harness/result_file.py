"""`result/Initialization_result.txt` in the reference's format (fileout_calib_result, src/laserMapping.cpp:708-725) — node I/O
that stays outside the accelerated path; here for the plumbing test of BASELINE.json configs[0].
Eigen's default stream format: fixed 6 decimals (the stream is set to fixed / setprecision(6)), coefficients of one matrix
right-aligned to the widest one, separated by one space, rows on their own lines."""
from __future__ import annotations

import numpy as np


def _eigen(m):
    m = np.atleast_2d(np.asarray(m, float))
    cells = [[f"{v:.6f}" for v in row] for row in m]
    w = max(len(c) for row in cells for c in row)
    return "\n".join(" ".join(c.rjust(w) for c in row) for row in cells)


def rot_to_euler_deg(R):
    """RotMtoEuler (include/so3_math.h:109-129) times 57.3 as the reference prints it."""
    sy = np.sqrt(R[0, 0] * R[0, 0] + R[1, 0] * R[1, 0])
    if sy >= 1e-6:
        e = np.array([np.arctan2(R[2, 1], R[2, 2]), np.arctan2(-R[2, 0], sy), np.arctan2(R[1, 0], R[0, 0])])
    else:
        e = np.array([np.arctan2(-R[1, 2], R[1, 1]), np.arctan2(-R[2, 0], sy), 0.0])
    return e * 57.3


def write_result(path, title, R_LI, T_LI, time_lag, bias_g, bias_a, gravity, append=False):
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = R_LI, T_LI
    with open(path, "a" if append else "w") as f:
        f.write(f"{title}\n")
        f.write(f"Rotation LiDAR to IMU (degree)     = {_eigen(rot_to_euler_deg(np.asarray(R_LI)))}\n")
        f.write(f"Translation LiDAR to IMU (meter)   = {_eigen(T_LI)}\n")
        f.write(f"Time Lag IMU to LiDAR (second)     = {time_lag:.6f}\n")
        f.write(f"Bias of Gyroscope  (rad/s)         = {_eigen(bias_g)}\n")
        f.write(f"Bias of Accelerometer (meters/s^2) = {_eigen(bias_a)}\n")
        f.write(f"Gravity in World Frame(meters/s^2) = {_eigen(gravity)}\n\n")
        f.write("Homogeneous Transformation Matrix from LiDAR to IMU: \n")
        f.write(_eigen(T) + "\n\n\n")


def parse_result(path):
    """-> list of dicts (one per block: 'Initialization result:' / 'Refinement result:')."""
    blocks, cur = [], None
    lines = open(path).read().split("\n")
    i = 0
    while i < len(lines):
        ln = lines[i]
        if ln.endswith("result:"):
            cur = {"title": ln}
            blocks.append(cur)
        elif "=" in ln and cur is not None:
            k, v = ln.split("=")
            cur[k.strip()] = np.array([float(x) for x in v.split()])
        elif ln.startswith("Homogeneous") and cur is not None:
            cur["T"] = np.array([[float(x) for x in lines[i + 1 + r].split()] for r in range(4)])
            i += 4
        i += 1
    return blocks
