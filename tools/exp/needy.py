import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
import lidar_imu_init_amd as lii
from scipy.spatial import cKDTree
wl = bench.build_workload("stream100k", 8)
states0, tables = bench.start_states(wl)
n_full = max(len(s) for s in wl["scans"])
reg = lii.Registrar(max_scan_points=n_full + 1024, max_map_points=int(len(wl["map"]) * 1.5) + 1024, filter_size_map=wl["fs_map"])
reg.map_build(wl["map"]); reg.map_commit()
tree = cKDTree(wl["map"][:, :3].astype(np.float64))
cs = 3 * wl["fs_map"] if False else None
print("fs_map", wl["fs_map"], "fs_surf", wl["fs_surf"])
for j in (0, 1, 2):
    st = states0[j].copy()
    reg.scan_register(st, states0[j], imu_poses=tables[j], leaf=wl["fs_surf"], max_iterations=wl["max_it"], imu_en=True, scan_dev=reg.device_scan(wl["scans"][j]), scan_sorted=True)
    w = reg.scan_download(2)[:, :3].astype(np.float64)
    d, _ = tree.query(w, k=5)
    d5 = d[:, 4]
    for c in (0.45, 0.6, 0.75, 1.0):
        f = w / c - np.floor(w / c)
        mfrac = np.minimum(f, 1 - f).min(axis=1) * c
        guard = c + mfrac
        print(j, "cs", c, "needy (d5 > guard or > sqrt5)", int(np.sum(np.minimum(d5, np.sqrt(5.0)) > guard)), "d5>2.236", int(np.sum(d5 > np.sqrt(5.0))), "median d5", round(float(np.median(d5)), 3))
reg.close()
