"""One rank of a multi-rank registration job (run as a subprocess by tests/test_gpu_multirank.py).

argv: rank world uid_hex transport out.npz [device]
Every rank builds the same map, takes its contiguous block of the same scan, attaches the communicator and runs the
iterated update through lii_iekf_update and lii_iekf_iterate; the final state, the report and the 91 sums are saved.
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
    import lidar_imu_init_amd as lii
    from harness import synth
    from harness.lo_harness import so3_exp

    hall = synth.Hall(size=(24.0, 18.0, 6.0), n_boxes=8, seed=7)
    map_pts = hall.surface_points(0.15, noise=0.01, seed=7)
    reg = lii.Registrar(max_scan_points=40_000, max_map_points=400_000, filter_size_map=0.15, device=device)
    reg.map_build(map_pts)
    if world > 1:
        reg.comm_init(world, rank, uid, transport)
    states, reports, sums = [], [], []
    for k in range(n_scans):
        R = synth.rot_zyx(0.03, -0.02, 0.4 + 0.05 * k)
        p = np.array([0.8 + 0.1 * k, -0.6, 0.1])
        scan = synth.make_scan(hall, "vlp16", R, p, noise=0.02, seed=31 + k)
        lo, hi = (len(scan) * rank) // world, (len(scan) * (rank + 1)) // world
        st = lii.State()
        st.rot_end[:] = R @ so3_exp(np.array([0.003, -0.002, 0.004]))
        st.pos_end[:] = p + np.array([0.03, -0.02, 0.01])
        prop = st.copy()
        reg.scan_upload(scan[lo:hi])
        reg.downsample_skip()
        s91 = reg.iekf_iterate(st, True, True)  # at the common start state: only the summation order differs between worlds
        rep = reg.iekf_update(st, prop, max_iterations=5, imu_en=True)
        states.append(st.pod.copy())
        reports.append([rep["iterations"], rep["searches"], rep["effect_num"], int(rep["converged"])])
        sums.append(np.asarray(s91).copy())
    np.savez(out, states=np.array(states), reports=np.array(reports), sums=np.array(sums),
             transport=reg.comm_transport())
    reg.close()


if __name__ == "__main__":
    main()
