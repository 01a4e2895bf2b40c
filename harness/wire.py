"""Synthetic driver messages in the wire layouts the reference ingests (harness / test data, no computation of the path).

PointCloud2 point records follow the structs the reference registers with PCL (src/preprocess.h:35-116); the byte offsets
are the ones the real drivers publish (velodyne_pointcloud, ouster_ros, hesai/pandar, rslidar_sdk) and travel in the
`fields` tuple exactly as a ROS callback would read them from `msg->fields`:
    fields = (point_step, off_x, off_y, off_z, off_intensity, off_time, off_ring)
Livox CustomPoint (livox_ros_driver/CustomPoint.msg: uint32 offset_time, float32 x y z, uint8 reflectivity tag line) is
packed in the 20-byte in-memory layout of the generated C++ struct:
    fields = (point_step, off_offset_time, off_x, off_y, off_z, off_reflectivity, off_tag, off_line)
"""
from __future__ import annotations

import numpy as np

from . import synth

# LID_TYPE, reference include/common_lib.h:55
AVIA, VELO, OUSTER, L515, PANDAR, ROBOSENSE = 1, 2, 3, 4, 5, 6

DTYPES = {
    VELO: np.dtype({"names": ["x", "y", "z", "intensity", "ring", "time"],
                    "formats": ["<f4", "<f4", "<f4", "<f4", "<u2", "<f4"], "offsets": [0, 4, 8, 16, 20, 24], "itemsize": 32}),
    OUSTER: np.dtype({"names": ["x", "y", "z", "intensity", "t", "reflectivity", "ring", "ambient", "range"],
                      "formats": ["<f4", "<f4", "<f4", "<f4", "<u4", "<u2", "u1", "<u2", "<u4"],
                      "offsets": [0, 4, 8, 16, 20, 24, 26, 28, 32], "itemsize": 48}),
    PANDAR: np.dtype({"names": ["x", "y", "z", "intensity", "timestamp", "ring"],
                      "formats": ["<f4", "<f4", "<f4", "<f4", "<f8", "<u2"], "offsets": [0, 4, 8, 16, 24, 32], "itemsize": 48}),
    ROBOSENSE: np.dtype({"names": ["x", "y", "z", "intensity", "ring", "timestamp"],
                         "formats": ["<f4", "<f4", "<f4", "u1", "<u2", "<f8"], "offsets": [0, 4, 8, 12, 14, 16], "itemsize": 24}),
}
# Intel RealSense L515 through realsense-ros: pcl::PointXYZRGB (x y z f32, packed rgb at 16); the reference reads x, y, z only
DTYPES[L515] = np.dtype({"names": ["x", "y", "z", "rgb"], "formats": ["<f4", "<f4", "<f4", "<u4"], "offsets": [0, 4, 8, 16], "itemsize": 32})
LIVOX_DTYPE = np.dtype({"names": ["offset_time", "x", "y", "z", "reflectivity", "tag", "line"],
                        "formats": ["<u4", "<f4", "<f4", "<f4", "u1", "u1", "u1"], "offsets": [0, 4, 8, 12, 16, 17, 18],
                        "itemsize": 20})


def pc2_fields(lidar_type):
    d = DTYPES[lidar_type]
    if lidar_type == L515:  # no intensity / time / ring in the record: offsets of fields the handler never reads
        return (d.itemsize, 0, 4, 8, 0, 0, 0)
    tname = {VELO: "time", OUSTER: "t", PANDAR: "timestamp", ROBOSENSE: "timestamp"}[lidar_type]
    off = {n: d.fields[n][1] for n in d.names}
    return (d.itemsize, off["x"], off["y"], off["z"], off["intensity"], off[tname], off["ring"])


def livox_fields():
    off = {n: LIVOX_DTYPE.fields[n][1] for n in LIVOX_DTYPE.names}
    return (LIVOX_DTYPE.itemsize, off["offset_time"], off["x"], off["y"], off["z"], off["reflectivity"], off["tag"], off["line"])


def raw_sweep(hall: synth.Hall, sensor: str, R_wb, p_wb, noise=0.02, seed=7, max_range=100.0, nan_fraction=0.01):
    """A FULL sweep in firing order (column-major: all rings of azimuth step 0, then step 1, ...), the way spinning-LiDAR
    drivers publish it: xyz (body frame), ring, time offset [ms].  Missing returns are (0,0,0) (Velodyne/Ouster style) and
    a few are NaN (organised clouds with is_dense = false)."""
    rings, cols, fd, fu = synth.SENSORS[sensor]
    rng = np.random.default_rng(seed)
    dirs_b, t_ms = synth.spinning_lidar(rings, cols, fd, fu)
    order = np.arange(rings * cols).reshape(rings, cols).T.reshape(-1)  # ring-major -> column-major
    dirs_b, t_ms = dirs_b[order], t_ms[order]
    ring = (order // cols).astype(np.int32)
    rngs = hall.raycast(np.asarray(p_wb, float), dirs_b @ np.asarray(R_wb).T)
    rngs = rngs + rng.normal(0, noise, len(rngs))
    ok = np.isfinite(rngs) & (rngs < max_range)
    xyz = np.where(ok[:, None], dirs_b * np.where(ok, rngs, 0.0)[:, None], 0.0).astype(np.float32)
    bad = rng.random(len(xyz)) < nan_fraction
    xyz[bad] = np.nan
    return xyz, ring, t_ms.astype(np.float64)


def pack_pcl2(lidar_type, xyz, ring, t_ms, stamp_s, with_time=True, seed=3):
    """Bytes of PointCloud2::data for one message.  with_time=False zeroes the per-point time (the drivers that do not
    provide it: the reference then synthesises it from the azimuth, src/preprocess.cpp:132-139,163-185)."""
    n = len(xyz)
    a = np.zeros(n, DTYPES[lidar_type])
    rng = np.random.default_rng(seed)
    a["x"], a["y"], a["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    if lidar_type == L515:
        a["rgb"] = rng.integers(0, 1 << 24, n)
        return a.tobytes()
    a["ring"] = ring
    if lidar_type == ROBOSENSE:
        a["intensity"] = rng.integers(0, 255, n)
    else:
        a["intensity"] = rng.uniform(0, 255, n).astype(np.float32)
    if lidar_type == VELO:
        a["time"] = (t_ms / 1000.0).astype(np.float32) if with_time else 0.0
    elif lidar_type == OUSTER:
        a["t"] = np.round(t_ms * 1e6).astype(np.uint32)
    elif lidar_type == PANDAR:
        a["timestamp"] = stamp_s + t_ms / 1000.0
    else:  # ROBOSENSE stamps the END of the sweep in the header (hence the reference's + 0.1 s)
        a["timestamp"] = (stamp_s - 0.1 + t_ms / 1000.0) if with_time else 0.0
    return a.tobytes()


def avia_sweep(hall: synth.Hall, R_wb, p_wb, n_points=24000, n_lines=6, noise=0.02, seed=11, sweep_ms=100.0):
    """A Livox-Avia-like message: non-repetitive rosette inside a 70.4 x 77.2 deg FoV, 6 lines interleaved, tags mostly
    0x10 / 0x00 with some other return types, a few exact duplicates (the reference drops consecutive equal points)."""
    rng = np.random.default_rng(seed)
    k = np.arange(n_points)
    t = k / n_points
    az = np.deg2rad(35.2) * np.sin(2 * np.pi * 17.0 * t) * np.cos(2 * np.pi * 3.1 * t)
    el = np.deg2rad(38.6) * np.sin(2 * np.pi * 23.0 * t + 0.3) + np.deg2rad(1.0) * ((k % n_lines) - 2.5)
    dirs = np.stack([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)], -1)
    rngs = hall.raycast(np.asarray(p_wb, float), dirs @ np.asarray(R_wb).T) + rng.normal(0, noise, n_points)
    ok = np.isfinite(rngs) & (rngs < 100.0)
    xyz = np.where(ok[:, None], dirs * np.where(ok, rngs, 0.0)[:, None], 0.0).astype(np.float32)
    a = np.zeros(n_points, LIVOX_DTYPE)
    dup = rng.random(n_points) < 0.01
    dup[0] = False
    for i in np.nonzero(dup)[0]:
        xyz[i] = xyz[i - 1]
    a["x"], a["y"], a["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    a["offset_time"] = np.round(t * sweep_ms * 1e6).astype(np.uint32)
    a["reflectivity"] = rng.integers(0, 255, n_points)
    a["line"] = k % n_lines
    a["line"][rng.random(n_points) < 0.005] = 7  # beyond N_SCANS
    tag = np.where(rng.random(n_points) < 0.9, 0x10, 0x00).astype(np.uint8)
    tag[rng.random(n_points) < 0.03] = 0x20
    tag |= rng.integers(0, 16, n_points).astype(np.uint8)  # low bits are unrelated flags
    a["tag"] = tag
    return a.tobytes(), n_points


def avia_message(hall: synth.Hall, traj, t_beg: float, sweep_s=0.1, n_points=24000, n_lines=6, noise=0.01, seed=11, max_range=100.0):
    """One Livox-Avia CustomMsg taken WHILE the platform moves along `traj` (ray k fired at t_beg + k / n * sweep_s from the pose
    of that instant): the rosette of avia_sweep, phase-continuous across messages so that the scan pattern does not repeat."""
    rng = np.random.default_rng(seed)
    k = np.arange(n_points)
    frac = k / n_points
    tj = t_beg + frac * sweep_s
    ph = tj / sweep_s  # pattern phase keeps running between messages (non-repetitive scanning)
    az = np.deg2rad(35.2) * np.sin(2 * np.pi * 17.0 * ph) * np.cos(2 * np.pi * 3.137 * ph)
    el = np.deg2rad(38.6) * np.sin(2 * np.pi * 23.0 * ph + 0.3) + np.deg2rad(1.0) * ((k % n_lines) - 2.5)
    dirs = np.stack([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)], -1)
    R = traj.R(tj)
    o = traj.p(tj)
    rngs = hall.raycast(o, np.einsum("nij,nj->ni", R, dirs)) + rng.normal(0, noise, n_points)
    ok = np.isfinite(rngs) & (rngs < max_range) & (rngs > 0.1)
    xyz = np.where(ok[:, None], dirs * np.where(ok, rngs, 0.0)[:, None], 0.0).astype(np.float32)
    a = np.zeros(n_points, LIVOX_DTYPE)
    a["x"], a["y"], a["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    a["offset_time"] = np.round(frac * sweep_s * 1e9).astype(np.uint32)
    a["reflectivity"] = rng.integers(0, 255, n_points)
    a["line"] = k % n_lines
    tag = np.where(rng.random(n_points) < 0.95, 0x10, 0x00).astype(np.uint8)
    tag[rng.random(n_points) < 0.02] = 0x20
    a["tag"] = tag
    return a.tobytes(), n_points
