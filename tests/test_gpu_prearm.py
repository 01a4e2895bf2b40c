"""The pre-armed prologue (lii_scan_job::next_scan_dev, ABI 8): the next scan's de-skew + filter-insert launch is
enqueued behind the current scan's passes and waits on the device for the record the next call writes.  Whatever happens to the
announcement - it is used, it names another scan, another entry point comes in between, nobody comes at all - the results are the
bits of the plain calls, and the stream never hangs.  Reference: the data dependency the wait honours is src/laserMapping.cpp:905-915
(p_imu->Process de-skews with the state the previous update left)."""
import os
import subprocess
import sys
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stream(n_scans=4):
    sys.path.insert(0, ROOT)
    import bench
    wl = bench.build_workload("vlp16", n_scans)
    states0, tables = bench.start_states(wl)
    return wl, states0, tables


def _run(reg, wl, states0, tables, dev, order, announce, between=None):
    """Registers the scans of `order`; announce(k) -> index of the scan announced in call k (or None).  Returns [(state pod, report)]."""
    out = []
    for k, j in enumerate(order):
        st = states0[j].copy()
        a = announce(k)
        rep = reg.scan_register(st, states0[j], imu_poses=tables[j], leaf=wl["fs_surf"], max_iterations=wl["max_it"], imu_en=True, scan_dev=dev[j],
                                scan_sorted=True, next_scan=None if a is None else dev[a])
        out.append((st.pod.copy(), rep, reg.scan_download(1).copy() if between == "download" else None))
        if between == "sync":
            reg.synchronize()
    return out


@pytest.mark.gpu
def test_an_announced_scan_is_registered_to_the_same_bits():
    import lidar_imu_init_amd as lii
    wl, states0, tables = _stream(4)
    n_full = max(len(s) for s in wl["scans"])
    reg = lii.Registrar(max_scan_points=n_full + 1024, max_map_points=int(len(wl["map"]) * 1.5) + 1024, filter_size_map=wl["fs_map"])
    reg.map_build(wl["map"])
    reg.map_commit()
    dev = [reg.device_scan(s) for s in wl["scans"]]
    order = [0, 1, 2, 3, 0, 1, 2, 3, 2, 1]
    plain = _run(reg, wl, states0, tables, dev, order, lambda k: None)
    # every call announces its successor: every prologue but the first is a pre-armed one
    right = _run(reg, wl, states0, tables, dev, order, lambda k: order[k + 1] if k + 1 < len(order) else None)
    # every call announces ANOTHER scan than the one that comes: the waiting launch is told to end, the scan is launched as always
    wrong = _run(reg, wl, states0, tables, dev, order, lambda k: order[k])
    # another entry point between the calls (a download / a synchronise): it ends the waiting launch first instead of waiting behind it
    t0 = time.perf_counter()
    dl = _run(reg, wl, states0, tables, dev, order, lambda k: order[k + 1] if k + 1 < len(order) else None, between="download")
    sy = _run(reg, wl, states0, tables, dev, order, lambda k: order[k + 1] if k + 1 < len(order) else None, between="sync")
    dt = time.perf_counter() - t0
    assert dt < 1.5, dt  # (a waiting launch ends itself after 2 s: nothing here waited for that)
    for name, got in (("announced", right), ("announced wrongly", wrong), ("download between", dl), ("synchronise between", sy)):
        for k, ((p0, r0, _), (p1, r1, _)) in enumerate(zip(plain, got)):
            assert r0["iterations"] == r1["iterations"] and r0["effect_num"] == r1["effect_num"], (name, k)
            assert np.array_equal(p0, p1), (name, k, np.abs(p0 - p1).max())
            assert np.array_equal(r0["normal_eq"], r1["normal_eq"]), (name, k)
    reg.close()


@pytest.mark.gpu
def test_a_launch_nobody_comes_for_ends_itself_and_the_stream_goes_on():
    """LII_PREARM_TIMEOUT_MS=20 in a child process: the announced scan never comes (the host sleeps past the bound), the waiting launch
    gives up by compare-and-swap on the host's state word, and the next call - which does bring the announced scan - finds the word
    EXPIRED and launches its prologue like any other: same bits, no hang."""
    code = r'''
import sys, time, numpy as np
sys.path.insert(0, %r)
import bench, lidar_imu_init_amd as lii
wl = bench.build_workload("vlp16", 2)
states0, tables = bench.start_states(wl)
n_full = max(len(s) for s in wl["scans"])
reg = lii.Registrar(max_scan_points=n_full + 1024, max_map_points=int(len(wl["map"]) * 1.5) + 1024, filter_size_map=wl["fs_map"])
reg.map_build(wl["map"]); reg.map_commit()
dev = [reg.device_scan(s) for s in wl["scans"]]
def call(j, nxt):
    st = states0[j].copy()
    rep = reg.scan_register(st, states0[j], imu_poses=tables[j], leaf=wl["fs_surf"], max_iterations=wl["max_it"], imu_en=True, scan_dev=dev[j], scan_sorted=True,
                            next_scan=None if nxt is None else dev[nxt])
    return st.pod.copy(), rep
a0, _ = call(0, None); a1, _ = call(1, None)
b0, _ = call(0, 1)
time.sleep(0.25)            # far beyond the 20 ms the waiting launch allows
t0 = time.perf_counter()
b1, _ = call(1, None)       # the announced scan comes - too late
dt = time.perf_counter() - t0
assert np.array_equal(a0, b0) and np.array_equal(a1, b1), "bits differ"
assert dt < 0.1, dt
reg.close()
print("OK")
''' % ROOT
    env = dict(os.environ, LII_PREARM_TIMEOUT_MS="20", LII_DIAG="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])
    assert "1 expired" in r.stderr, r.stderr[-800:]  # (LII_DIAG: "pre-armed prologues: 0 used, 0 cancelled, 1 expired")


@pytest.mark.gpu
def test_a_scan_announced_while_it_is_uploaded_gives_the_same_bits():
    """The complete pipeline (lii_scan_upload_next / lii_scan_register with the map update in the job / lii_scan_advance): the scan on
    its way is announced by the library itself and de-skewed where it lands.  The same loop with LII_PREARM=0 and 1, in two processes:
    every state bit for bit, and the pre-armed form did run."""
    code = r'''
import sys, hashlib, numpy as np
sys.path.insert(0, %r)
import bench, lidar_imu_init_amd as lii
wl = bench.build_workload("vlp16", 4)
states0, tables = bench.start_states(wl)
n_full = max(len(s) for s in wl["scans"])
reg = lii.Registrar(max_scan_points=n_full + 1024, max_map_points=int(len(wl["map"]) * 1.5) + 1024, filter_size_map=wl["fs_map"])
reg.map_build(wl["map"]); reg.map_commit()
hs = [np.ascontiguousarray(s) for s in wl["scans"]]
order = [0, 1, 2, 3, 1, 0, 3, 2, 2, 1]
reg.scan_upload_next(hs[order[0]]); reg.scan_advance()
m = hashlib.sha256()
for k, j in enumerate(order):
    if k + 1 < len(order):
        reg.scan_upload_next(hs[order[k + 1]])
    st = states0[j].copy()
    rep = reg.scan_register(st, states0[j], imu_poses=tables[j], leaf=wl["fs_surf"], max_iterations=wl["max_it"], imu_en=True, scan_sorted=True, map_update=True)
    m.update(st.pod.tobytes()); m.update(np.int64([rep["iterations"], rep["effect_num"]]).tobytes())
    if k + 1 < len(order):
        reg.scan_advance()
reg.synchronize()
print("HASH", m.hexdigest(), reg.map_size())
reg.close()
''' % ROOT
    out = {}
    for v in ("0", "1"):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, LII_PREARM=v, LII_DIAG="1"), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "HASH" in r.stdout, (v, r.stdout[-500:], r.stderr[-1500:])
        out[v] = (r.stdout.split("HASH")[1].split()[:2], r.stderr)
    assert out["0"][0] == out["1"][0], (out["0"][0], out["1"][0])
    import re
    used = int(re.search(r"pre-armed prologues: (\d+) used", out["1"][1]).group(1))
    assert used >= 8, out["1"][1][-600:]
    assert "pre-armed prologues: 0 used" in out["0"][1]


@pytest.mark.gpu
def test_a_job_of_abi_6_is_served_as_before():
    """A caller built against ABI 6 - 7 fills in the first 56 bytes of lii_scan_job and says so (struct_size 56): no announcement, same bits."""
    import ctypes as C
    import lidar_imu_init_amd as lii
    from lidar_imu_init_amd.api import lii_iekf_opts, lii_iekf_report, lii_scan_job
    wl, states0, tables = _stream(2)
    n_full = max(len(s) for s in wl["scans"])
    reg = lii.Registrar(max_scan_points=n_full + 1024, max_map_points=int(len(wl["map"]) * 1.5) + 1024, filter_size_map=wl["fs_map"])
    reg.map_build(wl["map"])
    reg.map_commit()
    dev = [reg.device_scan(s) for s in wl["scans"]]
    ref = _run(reg, wl, states0, tables, dev, [0, 1], lambda k: None)
    for j in (0, 1):
        job = lii_scan_job()
        job.struct_size = 56
        poses = np.ascontiguousarray(tables[j], np.float64).reshape(-1, 22)
        job.undistort, job.imu_poses, job.n_imu_poses = 1, poses.ctypes.data, len(poses)
        job.leaf = float(wl["fs_surf"])
        job.opts = lii_iekf_opts(int(wl["max_it"]), 1)
        job.scan_dev, job.n_scan_dev, job.scan_sorted = dev[j][0], dev[j][1], 1
        job.next_scan_dev, job.next_n_scan = dev[1 - j][0], dev[1 - j][1]  # (bytes behind the 56 the caller vouches for: must be ignored)
        st = states0[j].copy()
        rep = lii_iekf_report()
        rc = reg.L.lii_scan_register(reg.h, C.byref(job), st.pod.ctypes.data_as(C.c_void_p), states0[j].pod.ctypes.data_as(C.c_void_p), C.byref(rep))
        assert rc == 0
        assert np.array_equal(st.pod, ref[j][0]) and rep.iterations == ref[j][1]["iterations"]
    reg.close()
