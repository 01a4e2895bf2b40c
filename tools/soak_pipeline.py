"""Soak of the complete per-scan pipeline (GPU box): N steps of scan_upload_next / scan_register / map_incremental (no counts: predicted
sizes, own stream) / scan_advance on the cyclic bench stream; reports throughput, map size, the library's diagnostics and the drift of
the registered poses against the first cycle.  usage: python tools/soak_pipeline.py [steps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("LII_DIAG", "1")
import bench
import lidar_imu_init_amd as lii

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
wl = bench.build_workload("stream100k", 8)
states0, tables = bench.start_states(wl)
reg = lii.Registrar(max_scan_points=max(len(s) for s in wl["scans"]) + 1024, max_map_points=int(len(wl["map"]) * 1.5) + 1024, filter_size_map=wl["fs_map"])
reg.map_build(wl["map"]); reg.map_commit()
scans = [np.ascontiguousarray(s) for s in wl["scans"]]
first = {}
drift = 0.0
reg.scan_upload_next(scans[0]); reg.scan_advance()
t0 = time.perf_counter()
for k in range(steps):
    j = k % len(scans)
    if k + 1 < steps:
        reg.scan_upload_next(scans[(k + 1) % len(scans)])
    st = states0[j].copy()
    rep = reg.scan_register(st, states0[j], imu_poses=tables[j], leaf=float(wl["fs_surf"]), max_iterations=wl["max_it"], imu_en=True)
    reg.map_incremental(st, want_counts=(k == 0))
    if k + 1 < steps:
        reg.scan_advance()
    if j in first:
        drift = max(drift, float(np.linalg.norm(st.pos_end - first[j])))
    else:
        first[j] = st.pos_end.copy()
    assert rep["iterations"] >= 1
reg.synchronize()
dt = time.perf_counter() - t0
print(f"soak: {steps} steps, {steps / dt:.0f} scans/s (Python loop), map {reg.map_size()} points, max |p - p(first cycle)| = {drift:.2e} m")
reg.close()
