"""Search passes that leave MORE unfinished queries than the fit launch's completion workgroups take (kFlagCap = 256; a sensor that looks into
unmapped space): round 6 finishes the listed queries in a launch of its own between the search and the fit launch (k_complete_listed), which
the launch plan enqueues when the scan before listed that many.  The neighbour lists are exact on every path (ikd_Tree.cpp:825-968 semantics,
tests/test_gpu_full_size.py), and with every point fitted and summed by its own lane the result is the SAME BITS as with every workgroup
finishing its own (LII_WIDE_COMPLETION=0, the form of rounds 4 - 6) - whatever the plan predicted."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

HOLE = ((4.0, 16.0), (-6.0, 6.0), (-10.0, -1.0))  # a floor region missing from the map


def in_hole(w):
    (x0, x1), (y0, y1), (z0, z1) = HOLE
    return (w[:, 0] > x0) & (w[:, 0] < x1) & (w[:, 1] > y0) & (w[:, 1] < y1) & (w[:, 2] > z0) & (w[:, 2] < z1)


def test_listed_completion_gives_the_bits_of_every_workgroup_finishing_its_own(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    import lidar_imu_init_amd as lii
    wl = bench.build_workload("os1_128_cut3", 6)
    states0, tables = bench.start_states(wl)
    m = wl["map"]
    map_pts = np.ascontiguousarray(m[~in_hole(m)])
    # sub-frames 0 and 3 (the first third of two sweeps) look into the hole with thousands of points: as they are (more unfinished queries
    # than the list holds, 4096), with one in six of those points (hundreds to a few thousand: all listed), without them (a handful)
    full, mid, few = {}, {}, {}
    for j in (0, 3):
        sc = np.ascontiguousarray(wl["scans"][j])
        R, p = wl["poses"][j]
        w = sc[:, :3].astype(np.float64) @ np.asarray(R).T + np.asarray(p)
        hole = in_hole(w)
        keep6 = ~hole | (np.cumsum(hole) % 6 == 0)
        full[j], mid[j], few[j] = sc, np.ascontiguousarray(sc[keep6]), np.ascontiguousarray(sc[~hole])
    scans = {"full": full, "mid": mid, "few": few}
    # (two scans in a row switch the completion launch on or off: the stream holds runs of both kinds and single outliers)
    order = [("full", 0), ("mid", 3), ("mid", 0), ("few", 3), ("mid", 0), ("full", 3), ("few", 0), ("few", 3), ("mid", 0), ("few", 3), ("full", 0), ("mid", 3),
             ("full", 3)]

    def run(wide):
        monkeypatch.setenv("LII_WIDE_COMPLETION", "1" if wide else "0")
        r = lii.Registrar(max_scan_points=max(len(s) for s in full.values()) + 1024, max_map_points=int(len(map_pts) * 1.5) + 1024, filter_size_map=wl["fs_map"])
        out = []
        try:
            r.map_build(map_pts)
            r.map_commit()
            for kind, j in order:
                sc = scans[kind][j]
                st = states0[j].copy()
                r.scan_upload(sc)
                rep = r.scan_register(st, states0[j], imu_poses=tables[j], leaf=wl["fs_surf"], max_iterations=wl["max_it"], imu_en=True, scan_sorted=True)
                n_unf = r.last_unfinished_queries()
                n_d = len(r.scan_download(1))
                pts, cnt, sel = r.neighbors(n_d)
                out.append((st.pod.copy(), rep["iterations"], rep["searches"], rep["effect_num"], n_unf, pts.copy(), cnt.copy(), sel.copy()))
        finally:
            r.close()
        return out

    a = run(True)
    b = run(False)
    counts = [x[4] for x in a]
    assert counts == [x[4] for x in b]
    print("unfinished queries per scan (largest count among its search passes):", list(zip([k for k, _ in order], counts)))
    for (kind, _), c in zip(order, counts):
        assert (c > 256) == (kind != "few"), (order, counts)  # the stream crosses the completion workgroups' capacity in both directions
    assert any(c > 4096 for c in counts) and any(256 < c <= 4096 for c in counts), counts  # ... and the list's
    for k, (x, y) in enumerate(zip(a, b)):
        assert x[1:4] == y[1:4], (k, x[1:4], y[1:4])
        assert np.array_equal(x[6], y[6]) and np.array_equal(x[7], y[7]), k
        assert np.array_equal(x[5].view(np.uint32), y[5].view(np.uint32)), k
        assert np.array_equal(x[0], y[0]), f"scan {k}: the two forms registered to different states"
