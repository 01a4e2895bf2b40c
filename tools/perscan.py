"""Per-scan time of lii_scan_register on the eight distinct scans of the bench stream (median of 30 repetitions each, host clock around the
call): which scans cost more than the others - the one that looks past the edge of the map - and by how much.  LII_LIB selects the build.
usage (GPU box): python tools/perscan.py"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import lidar_imu_init_amd as lii
wl = bench.build_workload("stream100k", 8)
states0, tables = bench.start_states(wl)
n_full = max(len(s) for s in wl["scans"])
reg = lii.Registrar(max_scan_points=n_full + 1024, max_map_points=int(len(wl["map"]) * 1.5) + 1024, filter_size_map=wl["fs_map"])
reg.map_build(wl["map"]); reg.map_commit()
dev = [reg.device_scan(s) for s in wl["scans"]]
for rnd in range(3):
    for j in range(8):
        ts = []
        for r in range(30):
            st = states0[j].copy()
            reg.synchronize()
            t0 = time.perf_counter()
            rep = reg.scan_register(st, states0[j], imu_poses=tables[j], leaf=wl["fs_surf"], max_iterations=wl["max_it"], imu_en=True, scan_dev=dev[j], scan_sorted=True)
            reg.synchronize()
            ts.append(time.perf_counter() - t0)
        if rnd == 2:
            nd = len(reg.scan_download(1))
            nb, cnt, sel = reg.neighbors(nd)
            print(j, "us", round(1e6 * float(np.median(ts)), 1), "it", rep["iterations"], "searches", rep["searches"], "effect", rep["effect_num"], "n_down", nd, "lists<5", int((cnt < 5).sum()))
reg.close()
