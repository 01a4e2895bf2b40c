"""One rank of a multi-rank registration job (run as a subprocess by tests/test_gpu_multirank.py).

argv: rank world uid_hex transport out.npz [device]
Every rank builds the same map and receives the WHOLE scan (LII_WORKER_PARTITION=library, the default: the library splits the
down-sampled cloud by index; =voxel: by voxel, lii_comm_set_partition(h, 2)) or its contiguous block of it (=caller: the round-1
arrangement, no voxel filter), attaches the communicator
and runs (a) one host-driven pass at the common start state (lii_iekf_iterate: only the summation order differs between
worlds), (b) the whole per-scan call - de-skew, voxel filter ON, device-driven iterated update - through lii_scan_register,
(c) map_incremental.  The final state, the report, the 91 sums and the map size after every scan are saved.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world = int(sys.argv[1]), int(sys.argv[2])
    uid, transport, out = bytes.fromhex(sys.argv[3]), sys.argv[4], sys.argv[5]
    device = int(sys.argv[6]) if len(sys.argv) > 6 else 0
    n_scans = int(os.environ.get("LII_WORKER_SCANS", "3"))
    partition = os.environ.get("LII_WORKER_PARTITION", "library")
    caller_partition = partition == "caller"
    leaf = float(os.environ.get("LII_WORKER_LEAF", "0.1"))
    map_in_job = os.environ.get("LII_WORKER_MAP_IN_JOB") == "1"  # lii_scan_job::map_update instead of a call of lii_map_incremental
    import bench
    import lidar_imu_init_amd as lii
    from harness import synth
    from harness.lo_harness import so3_exp

    hall = synth.Hall(size=(24.0, 18.0, 6.0), n_boxes=8, seed=7)
    map_pts = hall.surface_points(0.15, noise=0.01, seed=7)
    reg = lii.Registrar(max_scan_points=40_000, max_map_points=400_000, filter_size_map=0.15, device=device)
    reg.map_build(map_pts)
    if world > 1 or transport == "rccl":
        reg.comm_init(world, rank, uid, transport)
        reg.comm_set_partition({"caller": 0, "library": 1, "voxel": 2}[partition])
    states, reports, sums, sums_b, map_sizes, n_down, n_local = [], [], [], [], [], [], []
    for k in range(n_scans):
        R = synth.rot_zyx(0.03, -0.02, 0.4 + 0.05 * k)
        p = np.array([0.8 + 0.1 * k, -0.6, 0.1])
        scan = synth.make_scan(hall, "vlp16", R, p, noise=0.02, seed=31 + k)
        scan = scan[np.argsort(scan[:, 3], kind="stable")]
        st = lii.State()
        st.rot_end[:] = R @ so3_exp(np.array([0.003, -0.002, 0.004]))
        st.pos_end[:] = p + np.array([0.03, -0.02, 0.01])
        prop = st.copy()
        if caller_partition:
            lo, hi = (len(scan) * rank) // world, (len(scan) * (rank + 1)) // world
            reg.scan_upload(scan[lo:hi])
            reg.downsample_skip()
            s91 = reg.iekf_iterate(st, True, True)
            rep = reg.iekf_update(st, prop, max_iterations=5, imu_en=True)
            n_down.append(hi - lo)
        else:
            table = bench.pose_table(prop.rot_end, prop.pos_end)
            reg.scan_upload(scan)
            reg.undistort_imu(table, prop.rot_end, prop.pos_end, prop.offset_R_L_I, prop.offset_T_L_I)
            nd, _ = reg.downsample(leaf)
            n_down.append(nd)
            s91 = reg.iekf_iterate(st, True, True)  # at the common start state
            rep = reg.scan_register(st, prop, imu_poses=table, leaf=leaf, max_iterations=5, imu_en=True, scan_dev=reg.device_scan(scan), scan_sorted=True,
                                    map_update=map_in_job)
            body_local = reg.scan_download(1)
            if k == 0:
                body0 = body_local.copy()
            n_local.append(len(body_local))  # what THIS rank holds of the cloud: its voxels (a split by voxel), or all of it
            # one more host-driven pass, now on the cloud lii_scan_register left behind (a split by voxel exists only there), at the
            # common start state: again only the summation order differs between worlds
            if not map_in_job and os.environ.get("LII_WORKER_NO_SUMS_B") != "1":  # (in the job: the pass would search the map the job has already updated)
                sums_b.append(np.asarray(reg.iekf_iterate(prop, True, True)).copy())
            if not map_in_job:
                reg.map_incremental(st)
            map_sizes.append(reg.map_size())
        states.append(st.pod.copy())
        reports.append([rep["iterations"], rep["searches"], rep["effect_num"], int(rep["converged"])])
        sums.append(np.asarray(s91).copy())
    np.savez(out, states=np.array(states), reports=np.array(reports), sums=np.array(sums), sums_b=np.array(sums_b), map_sizes=np.array(map_sizes),
             n_down=np.array(n_down), n_local=np.array(n_local), body0=body0 if not caller_partition and n_scans > 0 else np.zeros((0, 4), np.float32), describe=reg.comm_describe() if world > 1 else "", map_final=reg.map_download() if not caller_partition else np.zeros((0, 3), np.float32),
             transport=reg.comm_transport())
    reg.close()


if __name__ == "__main__":
    main()
