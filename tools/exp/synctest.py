import time, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
import lidar_imu_init_amd as lii
torch.cuda.init(); torch.cuda.synchronize()
def t(f, n=5):
    out = []
    for _ in range(n):
        a = time.perf_counter(); f(); out.append(round(1e6 * (time.perf_counter() - a), 1))
    return out
print("torch sync, nothing created:", t(torch.cuda.synchronize))
wl = bench.build_workload("vlp16", 2)
reg = lii.Registrar(max_scan_points=40000, max_map_points=400000, filter_size_map=wl["fs_map"])
print("after Registrar():", t(torch.cuda.synchronize))
reg.map_build(wl["map"]); reg.map_commit()
print("after map_build:", t(torch.cuda.synchronize))
states0, tables = bench.start_states(wl)
dev = [reg.device_scan(s) for s in wl["scans"]]
for k in range(30):
    st = states0[k % 2].copy()
    reg.scan_register(st, states0[k % 2], imu_poses=tables[k % 2], leaf=wl["fs_surf"], max_iterations=5, imu_en=True, scan_dev=dev[k % 2], scan_sorted=True)
reg.synchronize()
print("after 30 scans + lib sync:", t(torch.cuda.synchronize))
time.sleep(0.01)
print("after 10 ms idle:", t(torch.cuda.synchronize))
