"""The second half of BASELINE.json's metric - "extrinsic / time-offset error vs ref" - as a record for the bench line
(|dR| (deg), |dt| (m), |dt_d| (s), |dg|, |db| between the GPU evaluator and the oracle on the same buffers,
both against ground truth, and the fixture check against result/Initialization_result.txt).

Two inputs:
  * the reference's committed run (tests/golden/li_init/reference_run.npz <- Log/*.txt + result/Initialization_result.txt of
    the reference repository): lii_li_init_run (C++ conditioning chain + HIP residual / Jacobian evaluators + host LM) against the
    numpy oracle (oracle/li_init_np.py) on the same sequences, and against the numbers the reference program printed;
  * a synthetic LO -> LI-Init stream with KNOWN extrinsic, time offset, biases and gravity: LiDAR-only odometry on the GPU
    accumulates the LiDAR states, a simulated IMU runs alongside, then lii_li_init_run and the oracle solve the same buffers.
Harness code: it calls the library through lidar_imu_init_amd.api and the oracle as the checker (bench.py's parity / cpu_baseline
legs are where the oracle may be used; nothing here is timed into `value`).
Reference outputs: include/LI_init/LI_init.cpp:586-632 (LI_Initialization), result/Initialization_result.txt:1-13."""
from __future__ import annotations

import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _angle_deg(Ra, Rb):
    c = (np.trace(np.asarray(Ra).T @ np.asarray(Rb)) - 1.0) / 2.0
    return float(np.rad2deg(np.arccos(np.clip(c, -1.0, 1.0))))


def _gpu_result(res, total_lag):
    return dict(R_LI=np.array(res.R_LI[:]).reshape(3, 3), T_LI=np.array(res.T_LI[:]), time_lag=float(total_lag),
                gyro_bias=np.array(res.gyro_bias[:]), acc_bias=np.array(res.acc_bias[:]), grav=np.array(res.grav_L0[:]))


def _oracle_result(out):
    return dict(R_LI=out["stage2"]["R_LI"], T_LI=out["stage3"]["T_LI"], time_lag=float(out["time_delay"]),
                gyro_bias=out["stage2"]["gyro_bias"], acc_bias=out["stage3"]["acc_bias"], grav=out["stage3"]["grav_L0"])


def delta(a, b, with_lag=True):
    """|dR| as the angle of R_a^T R_b (deg), the rest as Euclidean norms."""
    d = dict(dR_deg=_angle_deg(a["R_LI"], b["R_LI"]), dt_m=float(np.linalg.norm(a["T_LI"] - b["T_LI"])),
             dtd_s=abs(a["time_lag"] - b["time_lag"]) if with_lag else None, dg=float(np.linalg.norm(a["grav"] - b["grav"])),
             dbg=float(np.linalg.norm(a["gyro_bias"] - b["gyro_bias"])), dba=float(np.linalg.norm(a["acc_bias"] - b["acc_bias"])))
    return d


def _seq_from_records(rec):
    from oracle import li_init_np as LI
    rec = np.asarray(rec, np.float64).reshape(-1, 22)
    s = LI.CalibSeq(len(rec))
    s.rot_end = rec[:, 0:9].reshape(-1, 3, 3).copy()
    s.ang_vel, s.linear_vel, s.ang_acc, s.linear_acc, s.t = (rec[:, 9:12].copy(), rec[:, 12:15].copy(), rec[:, 15:18].copy(),
                                                               rec[:, 18:21].copy(), rec[:, 21].copy())
    return s


def fixture_record(reg):
    """The reference's committed run."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import li_init_fixture as F
    from harness import synth
    d = F.load()
    imu, lid = F.sequences()
    t0 = time.perf_counter()
    res, lag1, total = reg.li_init_run(imu.to_records(), lid.to_records(), 10, 5)  # avia.yaml: orig_odom_freq 10, cut_frame_num 5
    t_gpu = time.perf_counter() - t0
    t0 = time.perf_counter()
    out = F.run(solve=True)
    t_cpu = time.perf_counter() - t0
    g, o = _gpu_result(res, total), _oracle_result(out)
    e = d["result_rot_euler_deg"] / 57.3  # the reference prints RotMtoEuler x 57.3 (src/laserMapping.cpp:711)
    ref = dict(R_LI=synth.rot_zyx(e[0], e[1], e[2]), T_LI=d["result_trans"], time_lag=0.0, gyro_bias=d["result_gyro_bias"],
               acc_bias=d["result_acc_bias"], grav=d["result_gravity"])
    return {"input": "reference's committed run (Log/*.txt + result/Initialization_result.txt -> tests/golden/li_init/reference_run.npz), "
                     f"{len(imu)} states",
            "gpu_vs_oracle": delta(g, o), "gpu_vs_reference_result_file": delta(g, ref, with_lag=False),
            "oracle_vs_reference_result_file": delta(o, ref, with_lag=False),
            "note": "the result file's time lag is printed against the wall clock (1716257158.9 s) and cannot be compared; the lag of the run "
                    "(-0.08 s coarse) is pinned through Log/ in tests/test_gpu_calib.py.  The file's digits are those of Ceres stopping on "
                    "its relative-cost tolerance (SURVEY.md section 8c): 0.003 deg / 2.4e-5 rad/s / 0.55 mm is the expected distance",
            "lii_li_init_run_ms": 1e3 * t_gpu, "oracle_ms": 1e3 * t_cpu}


def synthetic_record(lii, n_scans=480):
    """LO on a synthetic stream -> LI_Initialization, ground truth known."""
    import ctypes as C
    from harness import synth
    from harness.lo_harness import LoOdometry
    from lidar_imu_init_amd import calib_state_array
    from oracle import li_init_np as LI
    hall = synth.Hall(size=(24.0, 18.0, 6.0), n_boxes=8, seed=7)
    traj = synth.Trajectory()
    sweep = 0.05  # 20 Hz sub-frames
    R_LI = synth.rot_zyx(np.deg2rad(-1.0), np.deg2rad(-0.3), np.deg2rad(88.0))
    T_LI = np.array([-0.02, 0.02, 0.17])
    b_g = np.array([0.002, 0.0007, -0.0004])
    b_a = np.array([0.006, -0.007, 0.008])
    t_off = 0.015
    t0 = time.perf_counter()
    reg = lii.Registrar(max_scan_points=20_000, max_map_points=600_000, filter_size_map=0.15)
    lo = LoOdometry(reg, filter_size_surf=0.1, max_iteration=5)
    for k in range(n_scans):
        lo.process(synth.make_distorted_scan(hall, "mid16k", traj, k * sweep, sweep, noise=0.01, seed=1000 + k), k * sweep)
    t_lo = time.perf_counter() - t0
    t_imu, gyro, accel = synth.simulate_imu(traj, -0.5, n_scans * sweep + 0.5, 200.0, R_LI, T_LI, b_g, b_a, t_off)
    imu_all = calib_state_array(len(t_imu))
    imu_all[:, 9:12], imu_all[:, 18:21], imu_all[:, 21] = gyro, accel, t_imu
    lid = lo.lidar_calib_states()
    L = lii.load_library()
    oi, ol = calib_state_array(len(lid)), calib_state_array(len(lid))
    n = C.c_int32(0)
    rc = L.lii_li_init_interpolate(imu_all.ctypes.data_as(C.c_void_p), len(imu_all), lid.ctypes.data_as(C.c_void_p), len(lid), 2.5,
                                   oi.ctypes.data_as(C.c_void_p), ol.ctypes.data_as(C.c_void_p), C.byref(n))
    if rc != 0:
        raise RuntimeError(f"lii_li_init_interpolate: status {rc}")
    t0 = time.perf_counter()
    res, lag1, total = reg.li_init_run(oi[:n.value], ol[:n.value], 20, 1)
    t_gpu = time.perf_counter() - t0
    out = LI.li_initialization(_seq_from_records(oi[:n.value]), _seq_from_records(ol[:n.value]), 20, 1, solve=True)
    reg.close()
    g, o = _gpu_result(res, total), _oracle_result(out)
    # the CV odometry reports the MEAN rate of the last sub-frame at its end stamp, i.e. a signal delayed by sweep / 2: the lag the
    # method can observe is the IMU clock offset minus that delay (the reference's LO has the same property)
    truth = dict(R_LI=R_LI, T_LI=T_LI, time_lag=t_off - sweep / 2, gyro_bias=b_g, acc_bias=b_a, grav=np.array([0.0, 0.0, -9.81]))
    return {"input": f"synthetic LO -> LI-Init stream: {n_scans} sub-frames of ~16 k points at 20 Hz through the GPU odometry, IMU 200 Hz, "
                     f"{n.value} aligned states; truth: Euler (-1, -0.3, 88) deg, T (-0.02, 0.02, 0.17) m, IMU clock offset 15 ms "
                     "(observable: 15 ms - sweep / 2 = -10 ms), b_g 2e-3 rad/s, b_a within the solver's +-0.01 bound, |g| 9.81",
            "gpu_vs_oracle": delta(g, o), "gpu_vs_truth": delta(g, truth), "oracle_vs_truth": delta(o, truth),
            "note": "errors against the truth are those of the METHOD on a 16 k-point sensor (translation and accelerometer bias are its weakest "
                    "observables; the gravity vector is expressed in the first LiDAR frame, its distance to (0, 0, -9.81) includes that frame's "
                    "tilt), not of the arithmetic - gpu_vs_oracle is the arithmetic",
            "odometry_s": t_lo, "lii_li_init_run_ms": 1e3 * t_gpu}


def calibration_record(lii, reg, synthetic=True):
    rec = {"reference_outputs": "include/LI_init/LI_init.cpp:586-632, result/Initialization_result.txt:1-13",
           "units": {"dR_deg": "deg", "dt_m": "m", "dtd_s": "s", "dg": "m/s^2", "dbg": "rad/s", "dba": "m/s^2"}}
    rec["fixture"] = fixture_record(reg)
    if synthetic:
        rec["synthetic"] = synthetic_record(lii)
    f = rec["fixture"]["gpu_vs_oracle"]
    rec.update({k: f[k] for k in ("dR_deg", "dt_m", "dtd_s", "dg", "dbg", "dba")})  # headline: GPU vs oracle on the reference's run
    return rec
