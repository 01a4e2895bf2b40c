import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
import lidar_imu_init_amd as lii
wl = bench.build_workload("stream100k", 8)
states0, tables = bench.start_states(wl)
n_full = max(len(s) for s in wl["scans"])
reg = lii.Registrar(max_scan_points=n_full + 1024, max_map_points=int(len(wl["map"]) * 1.5) + 1024, filter_size_map=wl["fs_map"])
reg.map_build(wl["map"]); reg.map_commit()
dev = [reg.device_scan(s) for s in wl["scans"]]
for j in (0, 1, 0, 1):
    for r in range(40):
        st = states0[j].copy()
        reg.set_profiling(1 if r == 10 else (3 if r > 10 else 0))
        if r >= 10:
            reg.set_profiling(3)
        rep = reg.scan_register(st, states0[j], imu_poses=tables[j], leaf=wl["fs_surf"], max_iterations=wl["max_it"], imu_en=True, scan_dev=dev[j], scan_sorted=True)
    reg.synchronize()
    kp, n = reg.kernel_profile()
    print(j, n, {k: round(1e3 * v[0] / max(v[1], 1), 1) for k, v in kp.items()})
    nd = len(reg.scan_download(1))
    nb, cnt, sel = reg.neighbors(nd)
    needy = reg  # placeholder
reg.close()
