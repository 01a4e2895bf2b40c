"""Shared loader: the reference's committed sample run (tests/golden/li_init/reference_run.npz) -> oracle sequences."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_cache = {}


def load():
    if "d" not in _cache:
        _cache["d"] = dict(np.load(os.path.join(HERE, "golden", "li_init", "reference_run.npz")))
    return _cache["d"]


def sequences():
    """IMU / LiDAR CalibSeq as dumped by fout_before_filter (the last element it omits is unavailable), with the
    LiDAR linear velocity and attitude taken from Log/mat_out.txt (6 significant digits)."""
    from harness import synth
    from oracle import li_init_np as LI
    d = load()
    ib, lb, mo = d["imu_before"], d["lidar_before"], d["mat_out"]
    n = len(ib)
    imu = LI.CalibSeq(n)
    imu.ang_vel, imu.linear_acc, imu.t = ib[:, 0:3].copy(), ib[:, 4:7].copy(), ib[:, 7].copy()
    lid = LI.CalibSeq(n)
    lid.ang_vel, lid.t = lb[:, 0:3].copy(), lb[:, 4].copy()
    lid.linear_vel = mo[:n, 6:9].copy()
    e = mo[:n, 0:3] / 57.3  # RotMtoEuler output printed x 57.3 (laserMapping.cpp:1162)
    lid.rot_end = np.stack([synth.rot_zyx(a, b, c) for a, b, c in e])
    return imu, lid


def run(solve=True):
    key = ("run", solve)
    if key not in _cache:
        from oracle import li_init_np as LI
        imu, lid = sequences()
        _cache[key] = LI.li_initialization(imu, lid, 10, 5, solve=solve)  # avia.yaml: orig_odom_freq 10, cut_frame_num 5
    return _cache[key]
