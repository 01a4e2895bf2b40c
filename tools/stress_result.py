#!/usr/bin/env python3
"""Stress of the result hand-off (device -> mapped host memory -> polling host) and of the pre-armed prologue: N registrations of the
eight scans of the bench stream from their fixed start states, every call announcing its successor; every result must be the bits the
same scan gave the first time (state incl. covariance, report).  usage: stress_result.py [N=20000]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import lidar_imu_init_amd as lii  # noqa: E402

n_calls = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
wl = bench.build_workload("vlp16", 8)
states0, tables = bench.start_states(wl)
n_full = max(len(s) for s in wl["scans"])
reg = lii.Registrar(max_scan_points=n_full + 1024, max_map_points=int(len(wl["map"]) * 1.5) + 1024, filter_size_map=wl["fs_map"])
reg.map_build(wl["map"])
reg.map_commit()
dev = [reg.device_scan(s) for s in wl["scans"]]
ref = {}
bad = 0
t0 = time.perf_counter()
for k in range(n_calls):
    j = k % 8
    st = states0[j].copy()
    rep = reg.scan_register(st, states0[j], imu_poses=tables[j], leaf=wl["fs_surf"], max_iterations=wl["max_it"], imu_en=True, scan_dev=dev[j], scan_sorted=True,
                            next_scan=dev[(k + 1) % 8])
    key = (st.pod.tobytes(), rep["iterations"], rep["effect_num"], rep["normal_eq"].tobytes())
    if j not in ref:
        ref[j] = key
    elif ref[j] != key:
        bad += 1
        if bad <= 5:
            print(f"call {k} (scan {j}): result differs from the first one: max |d state| {np.abs(np.frombuffer(key[0]) - np.frombuffer(ref[j][0])).max():.3e}", flush=True)
print(f"{n_calls} calls, {bad} results that differ, {1e6 * (time.perf_counter() - t0) / n_calls:.1f} us per call")
reg.close()
sys.exit(1 if bad else 0)
