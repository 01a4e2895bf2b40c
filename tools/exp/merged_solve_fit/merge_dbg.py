import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import bench
import test_gpu_launch_plan as T
wl = bench.build_workload("os1_128_cut3", 3)
states0, tables = bench.start_states(wl)
a = T._run({"LII_TEST": "no_merge"}, wl, states0, tables)
b = T._run({}, wl, states0, tables)
for k, (x, y) in enumerate(zip(a, b)):
    d = np.abs(x[0] - y[0])
    print(k, "iters", x[1], y[1], "searches", x[2], y[2], "effect", x[3], y[3], "max dstate", d[:36].max(), "dcov", d[36:].max(), "dne", np.abs(x[4] - y[4]).max() / np.abs(x[4]).max())
