import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, lidar_imu_init_amd as lii
wl = bench.build_workload("stream100k", 8)
reg = lii.Registrar(max_scan_points=max(len(s) for s in wl["scans"]) + 1024, max_map_points=int(len(wl["map"]) * 1.5) + 1024, filter_size_map=wl["fs_map"])
reg.map_build(wl["map"]); reg.map_commit()
states0, tables = bench.start_states(wl)
for k in range(24):
    j = k % 8
    reg.scan_upload(np.ascontiguousarray(wl["scans"][j]))
    st = states0[j].copy()
    reg.scan_register(st, states0[j], imu_poses=tables[j], leaf=float(wl["fs_surf"]), max_iterations=wl["max_it"], imu_en=True)
    t0 = time.perf_counter(); a, b = reg.map_incremental(st); reg.synchronize(); t1 = time.perf_counter()
    print(k, "n_add", a, "n_nodown", b, "map", reg.map_size(), "us", round((t1 - t0) * 1e6))
