"""Debug helper (GPU box): dump k-NN mismatches between the HIP path and the oracle for the unit-test world."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import lidar_imu_init_amd as lii
from harness import synth
from oracle import oracle as O
from conftest import make_state

hall = synth.Hall(size=(24.0, 18.0, 6.0), n_boxes=8, seed=7)
map_pts = hall.surface_points(0.15, noise=0.01, seed=7)
R = synth.rot_zyx(0.03, -0.02, 0.4); p = np.array([0.8, -0.6, 0.1])
scan = synth.make_scan(hall, sys.argv[1] if len(sys.argv) > 1 else "vlp16", R, p, noise=0.02, seed=3)
R_LI = synth.rot_zyx(0.01, 0.02, -0.015); T_LI = np.array([0.03, -0.02, 0.05])
Rw = R @ synth.rot_zyx(0.004, -0.003, 0.005) @ R_LI.T
pw = p + np.array([0.03, -0.02, 0.01]) - Rw @ T_LI
st = make_state(O, Rw, pw, R_LI, T_LI)
tree = O.Tree("oracle"); tree.build(map_pts)
ref = tree.iterate_once(scan, st, search=True, imu_en=True, threads=4)
reg = lii.Registrar(max_scan_points=150_000, max_map_points=400_000, filter_size_map=0.15)
reg.map_build(map_pts); reg.scan_upload(scan); n = reg.downsample_skip()
out = reg.iekf_iterate(lii.State(st), True, True)
nb, cnt, sel = reg.neighbors(n)
world = reg.scan_download(2)
print("n", n, "count mismatch", (cnt != ref["nearest_n"]).sum())
bad = np.where((nb != ref["nearest"]).any(axis=(1, 2)) | (cnt != ref["nearest_n"]))[0]
print("mismatching queries", len(bad))
for i in bad[:8]:
    q = world[i, :3]
    dg = ((nb[i] - q) ** 2).sum(1); dr = ((ref["nearest"][i] - q) ** 2).sum(1)
    print("query", i, q, "cnt gpu/ref", cnt[i], ref["nearest_n"][i])
    print("  gpu d2", dg, "\n  ref d2", dr)
    print("  gpu pts", nb[i].tolist(), "\n  ref pts", ref["nearest"][i].tolist())
print("selected xor", np.logical_xor(sel, ref["selected"]).sum(), "effect gpu/ref", out[90], ref["out91"][90])
print("rel err out91", np.max(np.abs(out[:90] - ref["out91"][:90])) / np.max(np.abs(ref["out91"][:90])))
