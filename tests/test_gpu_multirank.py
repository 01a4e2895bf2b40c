"""Two ranks of one job on the box's single GPU: the node-local mailbox transports (lii_comm_init).

RCCL refuses two ranks on one device, the mailbox does not care which device a rank drives, so the whole multi-rank
path - rendezvous in the shared segment, the exchange inside k_reduce_solve, the device-side schedule staying in
lock-step - runs here exactly as it does with one GPU per rank: in its peer-mapped HBM form (slots in fine-grained device
memory, exported and opened through HIP IPC handles - between two processes on ONE device here, over xGMI on a multi-GPU
node) and in its host-memory form.  Checked:
  * both ranks end every scan with the BIT-identical state, report and sums (the sum is formed in rank order on each);
  * they agree with the single-rank run within the re-association of an fp64 sum (1e-11 relative on the sums of the
    first pass, 1e-6 m / rad on the final pose as everywhere in this suite) - WITH the voxel filter on: every rank hands over
    the whole scan, the de-skew and the filter run replicated and the library splits the down-sampled cloud (SURVEY.md
    section 8e: a voxel is never split between ranks), and map_incremental keeps the replicated maps bit-identical;
  * the caller-partitioned arrangement (every rank hands over its own points) still works;
  * a rank that never shows up turns into LII_ERR_COMM on the other (no hang);
  * the RCCL transport (final sum / ncclAllReduce / solve as three launches) on a one-rank communicator - the only RCCL
    configuration a single-GPU box can execute - reproduces the fused single-GPU loop bit for bit.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _rows(a):
    """The map as a SET of points: the in-place update keeps the same points on every rank, not the same array order."""
    return np.unique(np.ascontiguousarray(a, np.float32), axis=0)


def _n_devices():
    import ctypes
    import lidar_imu_init_amd as lii
    n = ctypes.c_int(0)
    assert lii.load_library().lii_device_count(ctypes.byref(n)) == 0
    return n.value


def _run_ranks(tmp_path, world, transport="auto", timeout=300, env=None, devices=None):
    import lidar_imu_init_amd as lii
    r = lii.Registrar(max_scan_points=1024, max_map_points=1024)
    uid = r.comm_unique_id().hex()
    r.close()
    procs, outs = [], []
    _run_ranks.calls = getattr(_run_ranks, "calls", 0) + 1
    for rank in range(world):
        out = str(tmp_path / f"job{_run_ranks.calls}_w{world}_r{rank}.npz")
        outs.append(out)
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "rank_worker.py"), str(rank), str(world), uid,
                                       transport, out] + ([str(devices[rank])] if devices else []),
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=dict(os.environ, **(env or {}))))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
        logs.append(o.decode(errors="replace"))
    for p, log in zip(procs, logs):
        assert p.returncode == 0, log[-3000:]
    return [dict(np.load(o)) for o in outs]  # (read now: the archive is opened lazily)


@pytest.mark.parametrize("transport", ["mailbox", "mailbox_host"])
def test_two_ranks_meet_in_the_mailbox(tmp_path, transport):
    one = _run_ranks(tmp_path, 1)[0]
    two = _run_ranks(tmp_path, 2, transport=transport)
    assert str(two[0]["transport"]) == transport and str(two[1]["transport"]) == transport
    for key in ("states", "reports", "sums"):
        assert np.array_equal(two[0][key], two[1][key]), key
    assert np.array_equal(one["reports"][:, [0, 1, 3]], two[0]["reports"][:, [0, 1, 3]])  # iterations, searches, converged
    assert np.all(np.abs(one["reports"][:, 2] - two[0]["reports"][:, 2]) <= 2)          # effect_num (1-ulp threshold flips)
    # sums at the common start state of the FIRST scan: same map, same cloud - only the summation order differs.  (From the
    # second scan on the maps of the two jobs may differ in a few points: a pose that differs by 1e-12 moves a float32 world
    # point by an ulp now and then, and map_incremental inserts it.)
    ref, got = one["sums"], two[0]["sums"]
    assert np.max(np.abs(ref[0] - got[0])) <= 1e-11 * np.max(np.abs(ref[0]))
    assert np.max(np.abs(ref - got)) <= 1e-6 * np.max(np.abs(ref))
    # lii_state: rot_end (9), pos_end (3) lead the POD
    # (a re-associated sum moves the iterate by ~1e-12, which can flip a point sitting on the plane / residual threshold in
    # a later pass: the suite-wide pose tolerance of tests/test_gpu_register.py applies, not the sum's)
    assert np.max(np.abs(one["states"][:, :12] - two[0]["states"][:, :12])) <= 1e-6
    # the voxel filter ran on the WHOLE scan on every rank: same down-sampled cloud as the single-rank job, and after
    # map_incremental the replicated maps are bit-identical to each other and equal to the single-rank map as a set
    assert np.array_equal(one["n_down"], two[0]["n_down"]) and np.array_equal(two[0]["n_down"], two[1]["n_down"])
    assert np.array_equal(two[0]["map_sizes"], two[1]["map_sizes"]) and np.array_equal(_rows(two[0]["map_final"]), _rows(two[1]["map_final"]))
    assert np.all(np.abs(one["map_sizes"] - two[0]["map_sizes"]) <= 3)  # (a pose that differs by 1e-12 can flip a keep-closest tie)


@pytest.mark.parametrize("transport", ["mailbox", "rccl", "mailbox_host"])
def test_two_ranks_on_two_devices(tmp_path, transport):
    """The same job with one DEVICE per rank - the arrangement the transports are made for: the HBM mailbox's remote stores cross
    xGMI (peer access checked and enabled at set-up, lii_mailbox.cpp), RCCL runs a real two-rank all-reduce.  Needs two visible
    devices: skipped on the single-GPU development boxes, enabled by itself on a multi-GPU node."""
    if _n_devices() < 2:
        pytest.skip("one visible device: the cross-device forms of the transports cannot run here")
    one = _run_ranks(tmp_path, 1)[0]
    two = _run_ranks(tmp_path, 2, transport=transport, devices=[0, 1])
    assert str(two[0]["transport"]) == transport and str(two[1]["transport"]) == transport
    for key in ("states", "reports", "sums"):
        assert np.array_equal(two[0][key], two[1][key]), key
    assert np.array_equal(one["reports"][:, [0, 1, 3]], two[0]["reports"][:, [0, 1, 3]])
    ref, got = one["sums"], two[0]["sums"]
    assert np.max(np.abs(ref[0] - got[0])) <= 1e-11 * np.max(np.abs(ref[0]))
    assert np.max(np.abs(one["states"][:, :12] - two[0]["states"][:, :12])) <= 1e-6
    assert np.array_equal(two[0]["map_sizes"], two[1]["map_sizes"]) and np.array_equal(_rows(two[0]["map_final"]), _rows(two[1]["map_final"]))


@pytest.mark.parametrize("transport", ["mailbox", "rccl"])
def test_two_ranks_split_by_voxel_on_two_devices(tmp_path, transport):
    """The split by voxel with one device per rank: the list exchange of the map update crosses xGMI (remote stores into the peer's
    gather area) or runs as two real ncclAllGather calls.  Skipped on the single-GPU development boxes."""
    if _n_devices() < 2:
        pytest.skip("one visible device: the cross-device forms of the transports cannot run here")
    env = {"LII_WORKER_PARTITION": "voxel"}
    one = _run_ranks(tmp_path, 1, env=env)[0]
    two = _run_ranks(tmp_path, 2, transport=transport, devices=[0, 1], env=env)
    assert all(str(t["transport"]) == transport and "split by voxel" in str(t["describe"]) for t in two), [str(t["describe"]) for t in two]
    for key in ("states", "reports", "sums", "sums_b", "map_sizes"):
        assert np.array_equal(two[0][key], two[1][key]), key
    assert np.array_equal(_rows(two[0]["map_final"]), _rows(two[1]["map_final"]))
    assert np.array_equal(two[0]["n_local"] + two[1]["n_local"], one["n_local"])
    assert np.max(np.abs(one["sums_b"][0] - two[0]["sums_b"][0])) <= 1e-11 * np.max(np.abs(one["sums_b"][0]))
    assert np.max(np.abs(one["states"][:, :12] - two[0]["states"][:, :12])) <= 1e-6
    assert np.all(np.abs(one["map_sizes"] - two[0]["map_sizes"]) <= 3)


def test_four_ranks_on_four_devices(tmp_path):
    if _n_devices() < 4:
        pytest.skip("fewer than four visible devices")
    four = _run_ranks(tmp_path, 4, devices=[0, 1, 2, 3])
    assert all(str(t["transport"]) == "mailbox" for t in four)
    for r in (1, 2, 3):
        for key in ("states", "reports", "sums", "map_sizes"):
            assert np.array_equal(four[0][key], four[r][key]), key


def test_three_ranks(tmp_path):
    three = _run_ranks(tmp_path, 3)  # "auto": the peer-mapped HBM mailbox is what three ranks on one node get
    assert all(str(t["transport"]) == "mailbox" for t in three)
    for r in (1, 2):
        for key in ("states", "reports", "sums", "map_sizes"):
            assert np.array_equal(three[0][key], three[r][key]), key
        assert np.array_equal(_rows(three[0]["map_final"]), _rows(three[r]["map_final"]))


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_split_the_cloud_by_voxel(tmp_path, world):
    """lii_comm_set_partition(h, 2) ("a voxel-key partition"): inside lii_scan_register every rank de-skews the
    scan, but inserts, filters, searches and fits only the voxels whose key hashes to it, and lii_map_incremental decides for those
    and EXCHANGES the two insert lists (lii_exchange.hip) instead of repeating the search for the whole cloud."""
    env = {"LII_WORKER_PARTITION": "voxel"}
    one = _run_ranks(tmp_path, 1, env=env)[0]
    many = _run_ranks(tmp_path, world, env=env)
    assert all(str(t["transport"]) == "mailbox" and "split by voxel" in str(t["describe"]) for t in many), [str(t["describe"]) for t in many]
    for r in range(1, world):
        for key in ("states", "reports", "sums", "sums_b", "map_sizes"):
            assert np.array_equal(many[0][key], many[r][key]), key
        assert np.array_equal(_rows(many[0]["map_final"]), _rows(many[r]["map_final"]))
    # the shares: disjoint and complete (they add up to the single-rank cloud), and even (a hash of the voxel key deals them out)
    shares = np.array([t["n_local"] for t in many])
    assert np.array_equal(shares.sum(axis=0), one["n_local"]), (shares, one["n_local"])
    assert np.all(np.abs(shares - one["n_local"] / world) <= 0.1 * one["n_local"] / world), shares
    # ... and they are the ones the published hash deals out (lidar_imu_init_amd/sharding.py restates it on the host): the centroid
    # of a voxel lies inside the voxel, so the single-rank cloud tells every voxel's key
    from harness import sharding
    owner = sharding.voxel_rank(sharding.voxel_keys(one["body0"][:, :3], float(os.environ.get("LII_WORKER_LEAF", "0.1"))), world)
    predicted = np.bincount(owner, minlength=world)
    assert np.all(np.abs(predicted - shares[:, 0]) <= 2), (predicted, shares[:, 0])  # (a centroid within an ulp of a voxel face)
    for r in range(world):
        assert len(many[r]["body0"]) == shares[r, 0]
    # the same 91 sums as the single-rank job up to the order of the additions - on the partitioned cloud (sums_b) as on the one
    # split by index (sums: the stand-alone filter in front of lii_iekf_iterate is not fused, hence not split by voxel)
    for key in ("sums", "sums_b"):
        ref, got = one[key], many[0][key]
        assert np.max(np.abs(ref[0] - got[0])) <= 1e-11 * np.max(np.abs(ref[0])), key
        assert np.max(np.abs(ref - got)) <= 1e-6 * np.max(np.abs(ref)), key
    assert np.array_equal(one["reports"][:, [0, 1, 3]], many[0]["reports"][:, [0, 1, 3]])
    assert np.all(np.abs(one["reports"][:, 2] - many[0]["reports"][:, 2]) <= 2)
    assert np.max(np.abs(one["states"][:, :12] - many[0]["states"][:, :12])) <= 1e-6
    # the exchanged lists hold the single-rank job's points in another order: the same map as a set (up to keep-closest ties / ulp flips)
    assert np.all(np.abs(one["map_sizes"] - many[0]["map_sizes"]) <= 3)
    # (the final poses of the two jobs differ by ~1e-9 - a threshold flip in a later pass - so a few hundred of the inserted world
    # points differ by a float ulp: every point of one map has its partner in the other within 1e-5 m, but for a handful)
    from scipy.spatial import cKDTree
    a, b = _rows(one["map_final"]), _rows(many[0]["map_final"])
    d_ab, d_ba = cKDTree(b).query(a)[0], cKDTree(a).query(b)[0]
    assert np.sum(d_ab > 1e-5) <= 5 and np.sum(d_ba > 1e-5) <= 5, (len(a), len(b), np.sum(d_ab > 1e-5), np.sum(d_ba > 1e-5))


@pytest.mark.parametrize("world", [1, 2])
def test_map_update_inside_the_job(tmp_path, world):
    """lii_scan_job::map_update = 1 instead of a call of lii_map_incremental: one rank enqueues the update behind the passes of the
    registration (map_update_early), a sharded job - its lists have to be exchanged first - makes it when the update has ended; the
    maps and the states are those of the explicit call either way."""
    env = {"LII_WORKER_PARTITION": "voxel"}
    a = _run_ranks(tmp_path, world, env=dict(env, LII_WORKER_NO_SUMS_B="1"))  # (no extra search pass between the update and its map update)
    b = _run_ranks(tmp_path, world, env=dict(env, LII_WORKER_MAP_IN_JOB="1"))
    for r in range(world):
        for key in ("states", "reports", "map_sizes", "n_local"):
            assert np.array_equal(a[r][key], b[r][key]), key
        assert np.array_equal(_rows(a[r]["map_final"]), _rows(b[r]["map_final"]))
    for r in range(1, world):
        assert np.array_equal(b[0]["states"], b[r]["states"]) and np.array_equal(_rows(b[0]["map_final"]), _rows(b[r]["map_final"]))


def test_list_exchange_equals_the_repeated_search(tmp_path):
    """A job split by index keeps its replicas identical in two ways: the insert lists of lii_map_incremental are exchanged (gather
    areas behind the mailbox slots), or - no gather areas: LII_TEST=no_gather here, the host-memory mailbox and RCCL in the field -
    every rank repeats the search for the whole cloud.  Same decisions, same lists, same order: bit-identical states, sums and map contents."""
    a = _run_ranks(tmp_path, 2, transport="mailbox")
    b = _run_ranks(tmp_path, 2, transport="mailbox", env={"LII_TEST": "no_gather"})
    assert "map lists exchanged" in str(a[0]["describe"]) and "repeats the search" in str(b[0]["describe"])
    for key in ("states", "reports", "sums", "sums_b", "map_sizes", "n_local"):
        assert np.array_equal(a[0][key], b[0][key]) and np.array_equal(a[1][key], b[1][key]), key
    assert np.array_equal(_rows(a[0]["map_final"]), _rows(b[0]["map_final"]))


@pytest.mark.parametrize("world", [2, 3, 5])
def test_list_exchange_layouts_with_several_ranks_on_one_device(world):
    """k_lists_push / k_lists_collect over the TRIMMED all-gather layout lists_exchange_rccl builds (block_bytes =
    header + the longest block of the job, every peers[] entry the same local area) had only ever run with one rank - a communicator
    holds one rank per device.  lii_selftest_list_exchange plays the ranks one after the other through the functions
    lists_exchange_rccl is made of, device copies standing in for the two ncclAllGather calls, and through the gather areas of the
    mailbox transport: both forms must hand every rank the lists of all ranks, concatenated in rank order, bit for bit."""
    import lidar_imu_init_amd as lii
    rng = np.random.default_rng(40 + world)
    reg = lii.Registrar(max_scan_points=6000, max_map_points=1000)
    try:
        for trial in range(3):
            # ragged on purpose: an empty rank, a rank with one list only, the longest block in the middle
            sizes_a = rng.integers(0, 900, world)
            sizes_n = rng.integers(0, 300, world)
            if trial == 0:
                sizes_a[0] = 0; sizes_n[0] = 0
            if trial == 1:
                sizes_a[-1] = 0
                sizes_n[world // 2] = 0
                sizes_a[world // 2] = 1100
            adds = [rng.normal(size=(int(k), 4)).astype(np.float32) for k in sizes_a]
            nods = [rng.normal(size=(int(k), 4)).astype(np.float32) for k in sizes_n]
            want_a, want_n = np.concatenate(adds), np.concatenate(nods)
            for form in ("gather", "allgather"):
                got_a, got_n = reg.selftest_list_exchange(adds, nods, form)
                assert got_a.shape == want_a.shape and got_n.shape == want_n.shape, (form, trial)
                assert got_a.tobytes() == want_a.tobytes() and got_n.tobytes() == want_n.tobytes(), (form, trial)
        # capacity: a rank's lists together may fill the scan, the joined lists may not exceed it
        with pytest.raises(lii.LIIError):
            reg.selftest_list_exchange([np.zeros((4000, 4), np.float32)] * 2, [np.zeros((0, 4), np.float32)] * 2, "allgather")
    finally:
        reg.close()


def test_one_process_rehearses_a_share(tmp_path):
    """LII_TEST=solo_share=<N> (tools/gpu_share.sh: the durations of ONE rank's launches without N devices): rank 0's share of an
    N-rank job split by voxel (N > 1) or by index (N < -1) in a single process, nothing exchanged."""
    from harness import sharding
    one = _run_ranks(tmp_path, 1)[0]
    by_voxel = _run_ranks(tmp_path, 1, env={"LII_TEST": "solo_share=4"})[0]
    by_index = _run_ranks(tmp_path, 1, env={"LII_TEST": "solo_share=-4"})[0]
    owner = sharding.voxel_rank(sharding.voxel_keys(one["body0"][:, :3], 0.1), 4)
    assert abs(int(by_voxel["n_local"][0]) - int(np.sum(owner == 0))) <= 2
    assert np.array_equal(by_index["n_local"], one["n_local"])  # (the whole cloud is there; a quarter of it is registered)
    for part in (by_voxel, by_index):  # a quarter of the effective points, a pose from a quarter of the rows: close to the full job's
        assert np.all(np.abs(part["reports"][:, 2] - one["reports"][:, 2] / 4) <= 0.1 * one["reports"][:, 2] / 4)
        assert np.max(np.abs(part["states"][:, :12] - one["states"][:, :12])) <= 5e-3


def test_caller_partitioned_ranks(tmp_path):
    """lii_comm_set_partition(0): every rank hands over its own block of an unfiltered scan (the round-1 arrangement)."""
    env = {"LII_WORKER_PARTITION": "caller"}
    one = _run_ranks(tmp_path, 1, env=env)[0]
    two = _run_ranks(tmp_path, 2, env=env)
    for key in ("states", "reports", "sums"):
        assert np.array_equal(two[0][key], two[1][key]), key
    assert np.max(np.abs(one["sums"] - two[0]["sums"])) <= 1e-11 * np.max(np.abs(one["sums"]))
    assert np.max(np.abs(one["states"][:, :12] - two[0]["states"][:, :12])) <= 1e-6


def test_rccl_transport_on_a_one_rank_communicator(tmp_path):
    """ncclCommInitRank(1 rank) + ncclAllReduce inside the loop: the three-launch form of the update must give the fused
    loop's result bit for bit (the all-reduce over one rank is the identity).  RCCL with N > 1 needs N devices and has not
    been executed anywhere yet (this box has one GPU) - stated in DESIGN.md section 6."""
    # (the map update of the RCCL job exchanges its insert lists - with itself here: pack, two ncclAllGather, collect - lii_capi_comm.cpp)
    fused = _run_ranks(tmp_path, 1)[0]
    rccl = _run_ranks(tmp_path, 1, transport="rccl")[0]
    assert str(rccl["transport"]) == "rccl" and str(fused["transport"]) == "none"
    for key in ("states", "reports", "sums", "map_sizes"):
        assert np.array_equal(fused[key], rccl[key]), key


def test_missing_rank_is_an_error_not_a_hang(tmp_path):
    import lidar_imu_init_amd as lii
    r = lii.Registrar(max_scan_points=1024, max_map_points=1024)
    uid = r.comm_unique_id()
    os.environ["LII_MAILBOX_TIMEOUT_S"] = "30,1.0"
    try:
        with pytest.raises(lii.LIIError):
            r.comm_init(2, 0, uid, "mailbox")  # the peer never arrives: the rendezvous gives up
        assert r.comm_transport() == "none"
    finally:
        del os.environ["LII_MAILBOX_TIMEOUT_S"]
        r.close()


def test_rank_that_stops_calling_times_out_on_the_device(tmp_path):
    """Rank 1 attaches and leaves without registering a scan; rank 0's reduce+solve kernel waits LII_MAILBOX_TIMEOUT_S for
    its flag, gives up, and the call returns LII_ERR_COMM."""
    import lidar_imu_init_amd as lii
    r = lii.Registrar(max_scan_points=1024, max_map_points=1024)
    uid = r.comm_unique_id().hex()
    r.close()
    procs = []
    for rank, scans in ((0, "1"), (1, "0")):
        env = dict(os.environ, LII_WORKER_SCANS=scans, LII_MAILBOX_TIMEOUT_S="1.0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "rank_worker.py"), str(rank), "2", uid, "mailbox",
                                       str(tmp_path / f"t{rank}.npz")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env))
    logs = [p.communicate(timeout=120)[0].decode(errors="replace") for p in procs]
    assert procs[1].returncode == 0, logs[1][-2000:]
    assert procs[0].returncode != 0 and "mailbox exchange timed out" in logs[0], logs[0][-2000:]
