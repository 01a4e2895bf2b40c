"""The dense cell window (GridView::win) through in-place map updates (round 6): whatever rewrites a cell entry or counts
an insert into it does the same to the window's copy, so a pipeline that updates the map behind every scan keeps searching through the
window.  Two processes run the same loop - scans registered with the map update in the job, the map growing - with LII_WINDOW_KEEP=1 and 0
(0: the first update drops the window, every later search walks the hashed tables): every state bit for bit, the map the same point set,
and the first run did keep its window.  (The neighbour lists behind a grown map are held to the unmodified reference tree in
tests/test_gpu_map.py, which now runs with the window kept as well.)"""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r'''
import sys, hashlib, numpy as np
sys.path.insert(0, %r)
import bench, lidar_imu_init_amd as lii
wl = bench.build_workload("vlp16", 6)
states0, tables = bench.start_states(wl)
n_full = max(len(s) for s in wl["scans"])
# the map starts WITHOUT the part of the hall the later scans see: the updates have something to add
m = wl["map"]
reg = lii.Registrar(max_scan_points=n_full + 1024, max_map_points=int(len(m) * 1.5) + 1024, filter_size_map=wl["fs_map"])
reg.map_build(m[(m[:, 0] < 6.0) | (m[:, 2] > 1.0)])
reg.map_commit()
dev = [reg.device_scan(s) for s in wl["scans"]]
h = hashlib.sha256()
for k in range(18):
    j = k %% 6
    st = states0[j].copy()
    rep = reg.scan_register(st, states0[j], imu_poses=tables[j], leaf=wl["fs_surf"], max_iterations=wl["max_it"], imu_en=True, scan_dev=dev[j], scan_sorted=True, map_update=True)
    h.update(st.pod.tobytes()); h.update(np.int64([rep["iterations"], rep["effect_num"], rep["searches"]]).tobytes())
pts = reg.map_download()
pts = pts[np.lexsort((pts[:, 2], pts[:, 1], pts[:, 0]))]
h.update(pts.tobytes())
print("HASH", h.hexdigest(), len(pts))
reg.close()
''' % ROOT


@pytest.mark.gpu
def test_the_window_kept_through_map_updates_gives_the_bits_of_the_hashed_tables():
    out = {}
    for keep in ("1", "0"):
        r = subprocess.run([sys.executable, "-c", CODE], env=dict(os.environ, LII_WINDOW_KEEP=keep, LII_DIAG="1"), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "HASH" in r.stdout, (keep, r.stdout[-500:], r.stderr[-1500:])
        m = re.search(r"dense cell window: kept current through (\d+) in-place updates, dropped (\d+) times", r.stderr)
        out[keep] = (r.stdout.split("HASH")[1].split()[:2], int(m.group(1)), int(m.group(2)))
    assert out["1"][0] == out["0"][0], out
    assert out["1"][1] >= 10, out["1"]          # the window lived through the updates ...
    assert out["0"][1] == 0 and out["0"][2] >= 1  # ... and LII_WINDOW_KEEP=0 dropped it at the first one
