"""Packs the reference's committed sample run (Log/*.txt, result/Initialization_result.txt — the only known-answer
data in the reference) into tests/golden/li_init/reference_run.npz.

Run once in the build container, where /root/reference exists; the GPU box only sees the .npz.
These are DATA files written by the reference program (LI_init.cpp:43-52, :135-156, :397-400, :478-485;
laserMapping.cpp:1162-1166), not source code.
"""
import os
import re

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "li_init", "reference_run.npz")


def main():
    L = lambda name: np.loadtxt(os.path.join(REF, "Log", name))
    imu_before = L("IMU_before_filter.txt")      # wx wy wz |w| ax ay az t          (1369 x 8)
    lidar_before = L("Lidar_before_filter.txt")  # wx wy wz |w| t                   (1369 x 5)
    imu_meas = L("IMU_meas.txt")                 # w(3) |w| a(3) alpha(3) t         (1331 x 11)
    lidar_meas = L("LiDAR_meas.txt")             # w(3) |w| (a_L - g)(3) alpha(3) t (1331 x 11)
    after_rot = L("Lidar_omg_after_rot.txt")     # (R w_L + b_g)(3) t               (1333 x 4)
    acc_cost = L("acc_cost.txt")                 # acc_I(3) acc_L(3) t_I t_L        (1333 x 8)
    mat_out = L("mat_out.txt")[:1371]            # LO rows: euler*57.3, pos, ext, ..., vel (12:15), omega (15:18)
    txt = open(os.path.join(REF, "result", "Initialization_result.txt")).read().split("Refinement result")[0]
    num = lambda key: np.array([float(x) for x in re.search(key + r"[^=]*=\s*(.*)", txt).group(1).split()])
    result = dict(rot_euler_deg=num("Rotation LiDAR to IMU"), trans=num("Translation LiDAR to IMU"),
                  gyro_bias=num("Bias of Gyroscope"), acc_bias=num("Bias of Accelerometer"), gravity=num("Gravity in World Frame"))
    np.savez_compressed(OUT, imu_before=imu_before, lidar_before=lidar_before, imu_meas=imu_meas, lidar_meas=lidar_meas,
                        after_rot=after_rot, acc_cost=acc_cost, mat_out=mat_out[:, [0, 1, 2, 3, 4, 5, 12, 13, 14, 15, 16, 17]],
                        **{"result_" + k: v for k, v in result.items()})
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
