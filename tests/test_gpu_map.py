"""The local map through the C-ABI: lii_map_build / lii_map_add_points must reproduce the point SET of the reference's
incremental k-d tree (Add_Points with per-voxel keep-closest-to-centre down-sampling, include/ikd-Tree/ikd_Tree.cpp:381-456)
and the device index built from it must answer exactly like Nearest_Search.  Also: a multi-scan stream
(register -> map_incremental -> next scan) stays on the oracle's trajectory."""
import os

import numpy as np
import pytest

from conftest import make_state

pytestmark = pytest.mark.gpu


def _as_set(a):
    return np.unique(np.ascontiguousarray(a, np.float32), axis=0)


def test_map_add_points_semantics(oracle):
    import lidar_imu_init_amd as lii
    rng = np.random.default_rng(3)
    ds = 0.5
    base = np.c_[rng.uniform(-20, 20, (40_000, 2)), rng.normal(0, 0.05, 40_000)].astype(np.float32)
    reg = lii.Registrar(max_scan_points=50_000, max_map_points=200_000, filter_size_map=ds)
    tree = oracle.Tree("oracle", downsample=ds)
    reg.map_build(base)
    tree.build(base)
    assert reg.map_size() == tree.validnum() == len(base)
    for s in range(4):
        add = (base[rng.choice(len(base), 6000)] + rng.normal(0, 0.3, (6000, 3))).astype(np.float32)
        # points exactly on voxel boundaries and exact duplicates exercise the float box predicate / same_point
        add[:50] = np.floor(add[:50] / ds).astype(np.float32) * np.float32(ds)
        add[50:60] = add[40:50]
        c_gpu = reg.map_add_points(add, True)
        c_ref = tree.add_points(add, True)
        assert c_gpu == c_ref
        plain = rng.uniform(-25, 25, (1500, 3)).astype(np.float32)
        reg.map_add_points(plain, False)
        tree.add_points(plain, False)
        assert reg.map_size() == tree.validnum()
    got, ref = _as_set(reg.map_download()), _as_set(tree.flatten())
    assert got.shape == ref.shape and np.array_equal(got, ref)
    # and the device index over that set answers like the tree
    q = (base[rng.choice(len(base), 20_000)] + rng.normal(0, 0.2, (20_000, 3))).astype(np.float32)
    scan = np.c_[q, np.zeros(len(q), np.float32)]
    reg.scan_upload(scan)
    n = reg.downsample_skip()
    reg.iekf_iterate(lii.State(oracle.state_init()), True, False)
    nb, cnt, _ = reg.neighbors(n)
    pts, d2, rc = tree.knn(q, threads=4)
    assert np.array_equal(cnt, rc)
    for k in range(5):
        m = cnt > k
        assert np.array_equal(nb[m, k], pts[m, k])
    reg.close()


def test_reference_tree_agrees_when_available(oracle):
    """Same stream against the UNMODIFIED reference ikd-Tree (oracle/_ref), when it has been built."""
    if not oracle.ref_available():
        pytest.skip("oracle/_ref not built on this box")
    import lidar_imu_init_amd as lii
    rng = np.random.default_rng(11)
    ds = 0.3
    base = np.c_[rng.uniform(-10, 10, (20_000, 2)), rng.normal(0, 0.03, 20_000)].astype(np.float32)
    reg = lii.Registrar(max_scan_points=30_000, max_map_points=100_000, filter_size_map=ds)
    ref = oracle.Tree("ref", downsample=ds)
    reg.map_build(base)
    ref.build(base)
    for s in range(3):
        add = (base[rng.choice(len(base), 3000)] + rng.normal(0, 0.2, (3000, 3))).astype(np.float32)
        assert reg.map_add_points(add, True) == ref.add_points(add, True)
    assert reg.map_size() == ref.validnum()
    q = (base[rng.choice(len(base), 5000)] + rng.normal(0, 0.1, (5000, 3))).astype(np.float32)
    reg.scan_upload(np.c_[q, np.zeros(len(q), np.float32)])
    n = reg.downsample_skip()
    reg.iekf_iterate(lii.State(oracle.state_init()), True, False)
    nb, cnt, _ = reg.neighbors(n)
    pts, d2, rc = ref.knn(q, threads=1)
    assert np.array_equal(cnt, rc)
    assert np.array_equal(nb[cnt == 5], pts[cnt == 5])
    reg.close()


def test_stream_with_map_incremental(oracle):
    """8 scans along a path: undistort (CV) -> voxel grid -> IEKF -> map_incremental, GPU pipeline vs oracle pipeline."""
    import lidar_imu_init_amd as lii
    from harness import synth
    hall = synth.Hall(size=(24.0, 18.0, 6.0), n_boxes=8, seed=7)
    fs_map, leaf = 0.3, 0.2
    reg = lii.Registrar(max_scan_points=40_000, max_map_points=400_000, filter_size_map=fs_map)
    tree = oracle.Tree("oracle", downsample=fs_map)
    st_g = make_state(oracle)
    st_o = st_g.copy()
    first = True
    for k in range(8):
        R = synth.rot_zyx(0.01 * k, -0.005 * k, 0.04 * k)
        p = np.array([0.15 * k, 0.05 * k, 0.0])
        scan = synth.make_scan(hall, "vlp16", R, p, noise=0.02, seed=50 + k)
        omega, vel = np.zeros(3), np.zeros(3)
        # --- oracle pipeline
        und_o = oracle.undistort_cv(scan, omega, vel, oracle.StateView(st_o).rot_end)
        body_o, _ = oracle.voxel_grid(und_o, leaf)
        # --- GPU pipeline
        reg.scan_upload(scan)
        reg.undistort_cv(omega, vel, lii.State(st_g).rot_end)
        nd, f = reg.downsample(leaf)
        body_g = reg.scan_download(1)
        # the oracle sorts by time before the voxel filter (float sums in a different order): same voxels, centroids within 1e-5
        assert nd == len(body_o)
        assert np.allclose(body_g[:, :3], body_o[:, :3], atol=2e-5)
        if first:
            # first scan: build the map from the world points (src/laserMapping.cpp:921-931)
            w = body_g[:, :3].astype(np.float64) @ lii.State(st_g).rot_end.T + lii.State(st_g).pos_end
            reg.map_build(w.astype(np.float32))
            tree.build(w.astype(np.float32))
            first = False
            continue
        # both pipelines now register the GPU's down-sampled cloud so that the comparison isolates the registration
        prop = oracle.state_boxplus(st_g, np.r_[0.002, -0.001, 0.04, 0.15, 0.05, 0.0, np.zeros(18)])  # motion prior
        sg = lii.State(prop)
        rep = reg.iekf_update(sg, lii.State(prop), max_iterations=4, imu_en=False)
        ro = tree.iekf_update(body_g, prop, prop, max_iterations=4, imu_en=False, threads=4)
        vo = oracle.StateView(ro["state"])
        assert rep["iterations"] == ro["iters"]
        assert np.linalg.norm(vo.pos_end - sg.pos_end) < 1e-6
        assert np.linalg.norm(oracle.log_so3(vo.rot_end.T @ sg.rot_end)) < 1e-7
        na, nn = reg.map_incremental(sg)
        a, b = tree.map_incremental(body_g, ro["state"], fs_map, apply=True)
        assert abs(na - len(a)) <= 2 and abs(nn - len(b)) <= 2   # 1-ulp pose differences can flip a voxel-centre test
        assert abs(reg.map_size() - tree.validnum()) <= 4
        st_g = sg.pod.copy()
        # sanity only: the estimate follows the simulated path (a 16-ring map built from ONE sweep is sparse between the
        # rings, so single-scan accuracy is a few cm for the reference algorithm as well — the oracle lands on the same pose)
        assert np.linalg.norm(sg.pos_end - p) < 0.10
    reg.close()


def test_delete_boxes_matches_the_reference_tree(oracle):
    """lii_map_delete_boxes against KD_TREE::Delete_Point_Boxes of the UNMODIFIED reference tree (min <= p < max on every axis,
    ikd_Tree.cpp:631): same count, same surviving set, and the rebuilt index answers like the tree afterwards."""
    import lidar_imu_init_amd as lii
    rng = np.random.default_rng(12)
    base = rng.uniform(-10, 10, (30_000, 3)).astype(np.float32)
    boxes = np.array([[-2, -2, -2, 2, 2, 2], [5, -10, -10, 10, 10, 0.5], [-9.5, 3, 3, -9.0, 3.5, 3.5], [20, 20, 20, 30, 30, 30]], np.float32)
    # points exactly on the faces: the lower face belongs to the box, the upper one does not
    base[:6] = [[-2, 0, 0], [2, 0, 0], [0, -2, 1.999], [0, 2, 0], [5, 0, 0.5], [5, 0, 0.4999]]
    reg = lii.Registrar(max_scan_points=20_000, max_map_points=60_000, filter_size_map=0.2)
    reg.map_build(base)
    inside = np.zeros(len(base), bool)
    for b in boxes:
        inside |= np.all((base >= b[:3]) & (base < b[3:]), axis=1)
    n_del = reg.map_delete_boxes(boxes)
    assert n_del == int(inside.sum()) and reg.map_size() == len(base) - n_del
    got = _as_set(reg.map_download())
    assert np.array_equal(got, _as_set(base[~inside]))
    if oracle.ref_available():
        tree = oracle.Tree("ref")
        tree.build(base)
        assert tree.delete_boxes(boxes) == n_del
        assert tree.validnum() == reg.map_size()
        assert np.array_equal(_as_set(tree.flatten()), got)
    # the index after the deletion: neighbours come from the survivors only
    q = rng.uniform(-3, 3, (4000, 3)).astype(np.float32)
    reg.scan_upload(np.c_[q, np.zeros(len(q), np.float32)])
    n = reg.downsample_skip()
    reg.iekf_iterate(lii.State(oracle.state_init()), True, False)
    nb, cnt, _ = reg.neighbors(n)
    t2 = oracle.Tree("oracle")
    t2.build(base[~inside])
    pts, d2, rc = t2.knn(q, threads=4)
    assert np.array_equal(cnt, rc)
    for k in range(5):
        m = cnt > k
        assert np.array_equal(nb[m, k], pts[m, k])
    assert reg.map_delete_boxes(np.zeros((0, 6), np.float32)) == 0 and reg.map_delete_boxes(boxes) == 0  # idempotent
    reg.close()


def test_capacity_overflow_leaves_the_map_untouched(oracle):
    """More points than max_map_points: LII_ERR_CAPACITY, nothing written outside the staging buffers (the appends drop writes
    at or beyond the capacity), the live map and its index exactly as before - lii_map_add_points with and without
    down-sampling, and lii_map_incremental."""
    import lidar_imu_init_amd as lii
    rng = np.random.default_rng(2)
    cap = 20_000
    base = rng.uniform(-8, 8, (18_000, 3)).astype(np.float32)
    reg = lii.Registrar(max_scan_points=8000, max_map_points=cap, filter_size_map=0.05)
    reg.map_build(base)
    before = reg.map_download().copy()
    q = rng.uniform(-8, 8, (3000, 3)).astype(np.float32)

    def knn_now():
        reg.scan_upload(np.c_[q, np.zeros(len(q), np.float32)])
        n = reg.downsample_skip()
        out = reg.iekf_iterate(lii.State(oracle.state_init()), True, False)
        return reg.neighbors(n), out

    (nb0, cnt0, sel0), out0 = knn_now()
    far = rng.uniform(20, 60, (6000, 3)).astype(np.float32)  # every point in a voxel of its own: 18 000 + 6 000 > 20 000
    for downsample in (False, True):
        with pytest.raises(lii.LIIError) as e:
            reg.map_add_points(far, downsample)
        assert e.value.code == -4
        assert reg.map_size() == len(base) and np.array_equal(reg.map_download(), before)
        (nb1, cnt1, sel1), out1 = knn_now()
        assert np.array_equal(nb0, nb1) and np.array_equal(cnt0, cnt1) and np.array_equal(out0, out1)
    # map_incremental: a scan far away from the map adds every point (no neighbour lists): overflow as well
    reg.scan_upload(np.c_[far, np.zeros(len(far), np.float32)])
    reg.downsample_skip()
    reg.iekf_iterate(lii.State(oracle.state_init()), True, False)
    with pytest.raises(lii.LIIError) as e:
        reg.map_incremental(lii.State(oracle.state_init()))
    assert e.value.code == -4
    assert reg.map_size() == len(base) and np.array_equal(reg.map_download(), before)
    # a batch that fits still goes in afterwards
    assert reg.map_add_points(far[:1500], False) == 0 and reg.map_size() == len(base) + 1500
    reg.close()


def test_update_that_runs_out_of_room_loses_nothing(oracle):
    """An in-place update provisions room for the usual batch only (a few new 8x8x8-cell blocks, some tail slots).  A batch that
    needs more - here forced with LII_TEST=map_tight: no spare block tables, a 256-slot tail - parks the inserts it cannot
    place; the next read of the counters rebuilds the index and inserts them again.  The caller sees no error and the map is
    the reference tree's point set (ikd_Tree.cpp:381-456) all the same - including batches far outside the mapped region
    (every point a new block) and searches right after."""
    import os
    import lidar_imu_init_amd as lii
    rng = np.random.default_rng(23)
    ds = 0.5
    base = np.c_[rng.uniform(-15, 15, (20_000, 2)), rng.normal(0, 0.05, 20_000)].astype(np.float32)
    os.environ["LII_TEST"] = "map_tight"
    try:
        reg = lii.Registrar(max_scan_points=30_000, max_map_points=120_000, filter_size_map=ds)
    finally:
        del os.environ["LII_TEST"]
    tree = oracle.Tree("oracle", downsample=ds)
    reg.map_build(base)
    tree.build(base)
    for s in range(5):
        # dense re-observation of the mapped area (slack runs out -> cells must move to the tail -> the tail runs out) ...
        add = (base[rng.choice(len(base), 5000)] + rng.normal(0, 0.4, (5000, 3))).astype(np.float32)
        assert reg.map_add_points(add, True) == tree.add_points(add, True)
        assert reg.map_size() == tree.validnum()
        # ... and a far-away patch: every point opens a new block
        far = (rng.uniform(-1, 1, (800, 3)) * 400 + np.array([3000.0 * (s + 1), 0, 0])).astype(np.float32)
        reg.map_add_points(far, False)
        tree.add_points(far, False)
        assert reg.map_size() == tree.validnum()
    got, ref = _as_set(reg.map_download()), _as_set(tree.flatten())
    assert got.shape == ref.shape and np.array_equal(got, ref)
    # an update that is NOT followed by a read of the counters (lii_map_incremental) and runs out of room, then a batch from the
    # host right behind it: the rebuild that completes the first must not eat the second
    scan = (base[rng.choice(len(base), 6000)] + rng.normal(0, 0.35, (6000, 3))).astype(np.float32)
    reg.scan_upload(np.c_[scan, np.zeros(len(scan), np.float32)])
    reg.downsample_skip()
    st = lii.State(oracle.state_init())
    reg.iekf_iterate(st, True, False)
    n_before = reg.map_size()
    na, nn = reg.map_incremental(st)
    batch = (rng.uniform(-1, 1, (700, 3)) * 50 + np.array([-5000.0, 0, 0])).astype(np.float32)
    reg.map_add_points(batch, False)
    got2 = _as_set(reg.map_download())
    assert len(_as_set(np.r_[got2, batch])) == len(got2)          # every point of the batch is in the map
    assert reg.map_size() == len(got2) and len(got2) >= n_before + len(batch)
    tree.add_points(batch, False)                                   # (the k-NN check below runs against base + far + batch)
    # (queries around the two far patches only: the tree did not take part in the map_incremental around `base`)
    q = np.r_[batch[rng.choice(len(batch), 500)] + rng.normal(0, 0.2, (500, 3)), far[:500] + rng.normal(0, 0.2, (500, 3))].astype(np.float32)
    reg.scan_upload(np.c_[q, np.zeros(len(q), np.float32)])
    n = reg.downsample_skip()
    reg.iekf_iterate(lii.State(oracle.state_init()), True, False)
    nb, cnt, _ = reg.neighbors(n)
    pts, d2, rc = tree.knn(q, threads=4)
    assert np.array_equal(cnt, rc)
    for k in range(5):
        m = cnt > k
        assert np.array_equal(nb[m, k], pts[m, k])
    reg.close()


def test_long_update_sequence_keeps_the_tree_set(oracle):
    """Thirty rounds of what a moving sensor does to the map - down-sampled re-observations of the mapped area, plain inserts,
    a patch of new ground ahead (new 8x8x8-cell blocks created inside the update), box deletions behind - applied IN PLACE:
    after every fifth round and at the end the device map is the tree's point set, `Add_Points` / `Delete_Point_Boxes` return the
    tree's counters every time, and the index answers like the tree (ikd_Tree.cpp:381-456, :500-516)."""
    import lidar_imu_init_amd as lii
    rng = np.random.default_rng(41)
    ds = 0.4
    base = np.c_[rng.uniform(-12, 12, (25_000, 2)), rng.normal(0, 0.05, 25_000)].astype(np.float32)
    reg = lii.Registrar(max_scan_points=30_000, max_map_points=400_000, filter_size_map=ds)
    tree = oracle.Tree("ref" if oracle.ref_available() else "oracle", downsample=ds)
    reg.map_build(base)
    tree.build(base)
    live = base
    for s in range(30):
        x0 = 12.0 + 4.0 * s  # the frontier moves along +x
        again = (live[rng.choice(len(live), 4000)] + rng.normal(0, 0.3, (4000, 3))).astype(np.float32)
        assert reg.map_add_points(again, True) == tree.add_points(again, True)
        ahead = np.c_[rng.uniform(x0, x0 + 4.0, 1500), rng.uniform(-12, 12, 1500), rng.normal(0, 0.05, 1500)].astype(np.float32)
        assert reg.map_add_points(ahead, True) == tree.add_points(ahead, True)
        plain = np.c_[rng.uniform(x0 - 8, x0 + 4, 300), rng.uniform(-12, 12, 300), rng.uniform(0.5, 3.0, 300)].astype(np.float32)
        reg.map_add_points(plain, False)
        tree.add_points(plain, False)
        if s % 3 == 2:  # drop a slab behind the sensor (lasermap_fov_segment's cub_needrm, src/laserMapping.cpp:260-305)
            xb = -12.0 + 4.0 * (s // 3)
            boxes = np.array([[xb, -13, -1, xb + 4.0, 13, 4]], np.float32)
            if tree.backend == "ref":
                assert reg.map_delete_boxes(boxes) == tree.delete_boxes(boxes)
            else:  # the restated tree has no box deletion of its own: rebuild it from the survivors
                pts_now = tree.flatten()
                inside = np.all((pts_now >= boxes[0, :3]) & (pts_now < boxes[0, 3:]), axis=1)
                assert reg.map_delete_boxes(boxes) == int(inside.sum())
                tree = oracle.Tree("oracle", downsample=ds)
                tree.build(pts_now[~inside])
        assert reg.map_size() == tree.validnum()
        if s % 5 == 4 or s == 29:
            got, ref = _as_set(reg.map_download()), _as_set(tree.flatten())
            assert got.shape == ref.shape and np.array_equal(got, ref), s
            live = got
    q = (live[rng.choice(len(live), 6000)] + rng.normal(0, 0.2, (6000, 3))).astype(np.float32)
    reg.scan_upload(np.c_[q, np.zeros(len(q), np.float32)])
    n = reg.downsample_skip()
    reg.iekf_iterate(lii.State(oracle.state_init()), True, False)
    nb, cnt, _ = reg.neighbors(n)
    pts, d2, rc = tree.knn(q, threads=4)
    assert np.array_equal(cnt, rc)
    for k in range(5):
        m = cnt > k
        assert np.array_equal(nb[m, k], pts[m, k])
    reg.close()


@pytest.mark.parametrize("hook", ["", "pred_small", "pred_small,force_rebuild", "force_rebuild", "job", "job,pred_small", "job,force_rebuild",
                                  "job,plan_force=0x10001", "job,force_rebuild,plan_force=0x10001", "emit_late", "job,emit_late"])
def test_map_incremental_without_counts_gives_the_same_map(hook):
    """lii_map_incremental with both size pointers NULL enqueues the update for PREDICTED list sizes on a stream of its own and
    returns at once; an update whose lists outgrow the prediction is repeated with the exact sizes before the next search
    (LII_TEST=pred_small: every one does), and folds the add list through a hash table instead of the batch sort (the box an
    Add_Points batch leaves behind does not depend on the batch order).  Same scans, same poses -> the same map, point for point,
    as the waiting, sorting form.  force_rebuild: every update takes the branch of a map low on room - the index is rebuilt first and
    the update runs on the handle's own stream; with pred_small on top every one of those updates outgrows its bounds and has to
    be repeated as well (that combination used to lose the scan's points silently).
    job: the map update rides in the registration job (lii_scan_job::map_update) - enqueued behind the update's passes before the
    host knows how the update ends; with plan_force=0x10001 (a launch plan that holds the first pass only) every update parks, the
    early launch sees that and does nothing, and the update is made when the loop has ended; with force_rebuild on top the index is
    rebuilt by the early enqueue of EVERY scan and the parked loop is continued behind it - its launches must see the rebuilt index,
    not the view taken before the passes went out.  emit_late: every seventh workgroup of k_map_decide (and of the voxel
    filter's emit) publishes its counts only when it is done - the workgroups above it decide its block again (prefix_below)."""
    import bench
    import lidar_imu_init_amd as lii
    wl = bench.build_workload("os1_128_cut3", 4)
    states0, tables = bench.start_states(wl)

    in_job = hook.startswith("job")
    hook = hook[4:] if in_job else hook

    def run(want_counts, env, in_job=False):
        old = os.environ.get("LII_TEST")
        if env:
            os.environ["LII_TEST"] = env
        try:
            reg = lii.Registrar(max_scan_points=140_000, max_map_points=1_600_000, filter_size_map=wl["fs_map"])
        finally:
            os.environ.pop("LII_TEST", None)
            if old is not None:
                os.environ["LII_TEST"] = old
        try:
            reg.map_build(wl["map"])
            out = []
            for rnd in range(2):
                for j, scan in enumerate(wl["scans"]):
                    st = states0[j].copy()
                    reg.scan_upload(scan)
                    rep = reg.scan_register(st, states0[j], imu_poses=tables[j], leaf=wl["fs_surf"], max_iterations=wl["max_it"], imu_en=True,
                                            map_update=in_job)
                    # the first call has nothing to predict from and waits either way
                    if not in_job:
                        reg.map_incremental(st, want_counts=want_counts or (rnd == 0 and j == 0))
                    out.append((st.pod.copy(), rep["iterations"], rep["effect_num"]))
            m = reg.map_download()
            return out, m[np.lexsort((m[:, 2], m[:, 1], m[:, 0]))]
        finally:
            reg.close()

    # the waiting form, folded through the batch sort (what lii_map_add_points does, and round 2 did here) - against the
    # returning form, folded through the hash table, on predicted sizes
    ref_out, ref_map = run(True, "fold_sort")
    got_out, got_map = run(False, hook, in_job)
    assert len(ref_map) > len(wl["map"])  # the map did grow
    for a, b in zip(ref_out, got_out):
        assert a[1] == b[1] and a[2] == b[2]
        assert np.array_equal(a[0], b[0])
    assert ref_map.shape == got_map.shape
    assert np.array_equal(ref_map, got_map)
