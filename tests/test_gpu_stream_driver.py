"""The C++ host loop bench.py times (harness/stream_driver.cpp) against the same calls made from Python: the final state and
the iteration totals of a short cyclic stream must be BIT-identical (both go through lii_scan_register with the scan handed
over as lii_scan_job::scan_dev)."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class StreamScan(C.Structure):
    _fields_ = [("scan_dev", C.c_void_p), ("n_points", C.c_int32), ("n_poses", C.c_int32), ("poses", C.c_void_p),
                ("state0", C.c_void_p)]


def test_cxx_loop_equals_python_loop(small_world):
    import lidar_imu_init_amd as lii
    from harness import synth
    from harness.lo_harness import so3_exp
    hall, map_pts = small_world
    drv = C.CDLL(os.path.join(ROOT, "harness", "libliinit_stream.so"))
    drv.lii_stream_run.restype = C.c_int
    drv.lii_stream_run.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_int32,
                                   C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    reg = lii.Registrar(max_scan_points=40_000, max_map_points=400_000, filter_size_map=0.15)
    reg.map_build(map_pts)
    scans, states, tables, devs = [], [], [], []
    for k in range(3):
        R = synth.rot_zyx(0.03, -0.02, 0.4 + 0.05 * k)
        p = np.array([0.8 + 0.1 * k, -0.6, 0.1])
        scan = synth.make_scan(hall, "vlp16", R, p, noise=0.02, seed=40 + k)
        scan[:, 3] = np.linspace(0, 100, len(scan), dtype=np.float32)
        st = lii.State()
        st.rot_end[:] = R @ so3_exp(np.array([0.003, -0.002, 0.004]))
        st.pos_end[:] = p + np.array([0.02, -0.02, 0.01])
        T = lii.pose6d_array(6)
        for j in range(6):
            T[j, 0] = 0.02 * j
            T[j, 4:7] = [1e-3, -2e-3, 1e-3]
            T[j, 7:10] = [1e-2, 0, 0]
            T[j, 10:13] = st.pos_end
            T[j, 13:22] = st.rot_end.reshape(-1)
        scans.append(scan)
        states.append(st)
        tables.append(np.ascontiguousarray(T, np.float64))
        devs.append(reg.device_scan(scan))
    steps, leaf, max_it = 7, 0.1, 5
    # Python loop
    it_py = se_py = 0
    last_py = None
    for k in range(steps):
        j = k % 3
        s = states[j].copy()
        rep = reg.scan_register(s, states[j], imu_poses=tables[j], leaf=leaf, max_iterations=max_it, imu_en=True, scan_dev=devs[j], scan_sorted=True)
        it_py += rep["iterations"]
        se_py += rep["searches"]
        last_py = s.pod.copy()
    # C++ loop
    stream = (StreamScan * 3)()
    for j in range(3):
        stream[j].scan_dev, stream[j].n_points = devs[j][0], devs[j][1]
        stream[j].poses, stream[j].n_poses = tables[j].ctypes.data, len(tables[j])
        stream[j].state0 = states[j].pod.ctypes.data
    totals = np.zeros(2, np.int64)
    last = lii.State()
    rc = drv.lii_stream_run(reg.h, C.byref(stream), 3, 0, steps, leaf, max_it, 1, 0, 0, totals.ctypes.data, last.pod.ctypes.data)
    assert rc == 0
    assert (int(totals[0]), int(totals[1])) == (it_py, se_py)
    assert np.array_equal(last.pod, last_py)
    # the same steps with an event in front of every launch (lii_set_profiling(h, 3), what bench.py's roofline.kernels is made
    # of): the same bits, and every kind of launch of the loop accounted for - one de-skew and one voxel launch per scan, one fit
    # and one solve per iteration, one k-NN and one search-pass fit per search
    totals2 = np.zeros(2, np.int64)
    last2 = lii.State()
    reg.set_profiling(1)
    rc = drv.lii_stream_run(reg.h, C.byref(stream), 3, 0, steps, leaf, max_it, 1, 0, -1, totals2.ctypes.data, last2.pod.ctypes.data)
    assert rc == 0
    reg.synchronize()
    kp, n_scans = reg.kernel_profile()
    reg.set_profiling(0)
    assert np.array_equal(last2.pod, last_py) and (int(totals2[0]), int(totals2[1])) == (it_py, se_py)
    assert n_scans == steps
    assert kp["deskew"][1] == steps and kp["voxel"][1] == steps
    assert kp["knn"][1] == se_py and kp["fit_search"][1] == se_py
    assert kp["fit_search"][1] + kp["fit"][1] == it_py and kp["solve"][1] == it_py
    for k, (ms, n) in kp.items():
        assert n == 0 or 1e-3 < ms / n < 5.0, (k, ms, n)  # between 1 us and 5 ms per launch
    reg.close()
