"""Stress of the map update's fused form (round 6: the fold's hash insert in k_map_decide, the inserts' cells found / created inside the fold
launch) against the six-launch form (LII_MAP_FUSE=0) on a map that GROWS: two handles register the same stream of scans taken while the
sensor walks through a hall of which the map at first knows one corner - every update creates blocks beside the fold's lookups.  The maps
are compared as point SETS every few steps (the map is a set, whatever the form), the registered states bit for bit.
usage: python tools/stress_mapfuse.py [steps] [sensor]      (GPU box; LII_TEST=pred_small adds the repeated-update path)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from harness import synth
import lidar_imu_init_amd as lii

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
sensor = sys.argv[2] if len(sys.argv) > 2 else "vlp16"
hall = synth.Hall(size=(60.0, 40.0, 8.0), n_boxes=20)
fs_map = 0.2


def pose(k):
    yaw = 0.05 * k
    return synth.rot_zyx(0.01 * np.sin(k), 0.01 * np.cos(k), yaw), np.array([-20.0 + 0.25 * k, 5.0 * np.sin(0.05 * k), 0.3])


def handle(fuse):
    os.environ["LII_MAP_FUSE"] = "1" if fuse else "0"
    r = lii.Registrar(max_scan_points=140_000, max_map_points=3_000_000, filter_size_map=fs_map)
    del os.environ["LII_MAP_FUSE"]
    return r


def as_set(m):
    m = np.ascontiguousarray(m[:, :3], np.float32)
    return m[np.lexsort((m[:, 2], m[:, 1], m[:, 0]))]


R0, p0 = pose(0)
first = synth.make_scan(hall, sensor, R0, p0, noise=0.01, seed=100)
world0 = (first[:, :3].astype(np.float64) @ R0.T + p0).astype(np.float32)
regs = [handle(True), handle(False)]
for r in regs:
    r.map_build(world0)
t0 = time.perf_counter()
worst = 0.0
for k in range(1, steps + 1):
    R, p = pose(k)
    scan = synth.make_scan(hall, sensor, R, p, noise=0.01, seed=100 + k)
    scan = scan[np.argsort(scan[:, 3], kind="stable")]
    pods = []
    for r in regs:
        st = lii.State()
        st.rot_end[:] = R
        st.pos_end[:] = p + np.array([0.01, -0.01, 0.005])
        s0 = st.copy()
        r.scan_upload(scan)
        rep = r.scan_register(st, s0, cv=True, leaf=0.1, max_iterations=4, imu_en=False, scan_sorted=True, map_update=True)
        pods.append((st.pod.copy(), rep["iterations"], rep["effect_num"]))
    assert pods[0][1:] == pods[1][1:], (k, pods[0][1:], pods[1][1:])
    assert np.array_equal(pods[0][0], pods[1][0]), f"step {k}: the two forms registered to different states"
    worst = max(worst, float(np.linalg.norm(pods[0][0][9:12] - p)))
    if k % 10 == 0 or k == steps:
        a, b = as_set(regs[0].map_download()), as_set(regs[1].map_download())
        assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"step {k}: the maps differ as sets ({len(a)} / {len(b)} points)"
print(f"stress_mapfuse: {steps} steps of {sensor} on a growing map, {regs[0].map_size()} points at the end, fused and six-launch forms: identical sets "
      f"and states; max |p - truth| {worst:.3f} m; {time.perf_counter() - t0:.1f} s")
for r in regs:
    r.close()
