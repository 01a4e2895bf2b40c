"""ctypes mirror of include/liinit_hip.h.

`Registrar` wraps one `lii_handle`.  Method names follow the reference's vocabulary
(src/laserMapping.cpp, src/IMU_Processing.hpp, include/LI_init/LI_init.h) so that tests read like the
reference's call sites:  map_build ~ ikdtree.Build, map_add_points ~ ikdtree.Add_Points,
undistort_imu/undistort_cv ~ ImuProcess::Process, downsample ~ downSizeFilterSurf.filter,
iekf_update ~ the iterated-Kalman loop of main(), map_incremental ~ map_incremental().
The shared library is mandatory: importing is cheap, but constructing a Registrar raises LIIError when
libliinit_hip.so is missing or no gfx950 device is usable (there is no CPU fallback by design).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (LII_LIB: another build of the same library - A/B measurements of tools/ab.sh; the default is the in-tree build)
_LIB = os.environ.get("LII_LIB") or os.path.join(_HERE, "lib", "libliinit_hip.so")

STATE_DOUBLES = 36 + 576


class LIIError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libliinit_hip error {code}: {msg}")
        self.code = code


class lii_config(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("device", C.c_int32), ("max_scan_points", C.c_int32),
                ("max_map_points", C.c_int32), ("map_cell_size", C.c_float), ("map_downsample_size", C.c_float),
                ("max_match_dist2", C.c_float), ("reserved0", C.c_float), ("plane_threshold", C.c_double),
                ("laser_point_cov_inv", C.c_double)]


class lii_iekf_opts(C.Structure):
    _fields_ = [("max_iterations", C.c_int32), ("imu_en", C.c_int32)]


class lii_iekf_report(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("searches", C.c_int32), ("effect_num", C.c_int32),
                ("converged", C.c_int32), ("normal_eq", C.c_double * 91)]


class lii_scan_job(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("undistort", C.c_int32), ("imu_poses", C.c_void_p),
                ("n_imu_poses", C.c_int32), ("leaf", C.c_float), ("opts", lii_iekf_opts), ("scan_dev", C.c_void_p),
                ("n_scan_dev", C.c_int32), ("scan_sorted", C.c_int32), ("map_update", C.c_int32),
                ("next_scan_dev", C.c_void_p), ("next_n_scan", C.c_int32), ("reserved1", C.c_int32),
                ("while_waiting", C.c_void_p), ("while_waiting_arg", C.c_void_p)]


WAIT_HOOK = C.CFUNCTYPE(None, C.c_void_p)  # lii_scan_job::while_waiting


class lii_kernel_profile(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("scans", C.c_int32), ("ms", C.c_double * 8), ("launches", C.c_int32 * 8)]


KERNEL_KINDS = ("deskew", "voxel", "knn", "fit_search", "fit", "solve")  # enum lii_kernel_kind


class lii_pc2_fields(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("point_step", "x", "y", "z", "intensity", "time", "ring")]


class lii_livox_fields(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("point_step", "offset_time", "x", "y", "z", "reflectivity", "tag", "line")]


class lii_ingest_opts(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("lidar_type", C.c_int32), ("n_scans", C.c_int32),
                ("point_filter_num", C.c_int32), ("blind", C.c_double), ("stamp_s", C.c_double),
                ("cut_frame_num", C.c_int32), ("scan_count", C.c_int32)]


class lii_frame_info(C.Structure):
    _fields_ = [("begin_time_s", C.c_double), ("last_offset_ms", C.c_double), ("offset", C.c_int32), ("count", C.c_int32)]


class lii_calib_result(C.Structure):
    _fields_ = [("R_LI", C.c_double * 9), ("T_LI", C.c_double * 3), ("gyro_bias", C.c_double * 3),
                ("acc_bias", C.c_double * 3), ("grav_L0", C.c_double * 3), ("time_lag_2", C.c_double),
                ("iterations", C.c_int32 * 3), ("final_cost", C.c_double * 3)]


_DECLS = {
    # name: (restype, argtypes)
    "lii_abi_version": (C.c_int, []),
    "lii_strerror": (C.c_char_p, [C.c_int]),
    "lii_last_error": (C.c_char_p, [C.c_void_p]),
    "lii_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "lii_create": (C.c_int, [C.POINTER(lii_config), C.POINTER(C.c_void_p)]),
    "lii_destroy": (C.c_int, [C.c_void_p]),
    "lii_synchronize": (C.c_int, [C.c_void_p]),
    "lii_map_reset": (C.c_int, [C.c_void_p]),
    "lii_map_build": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]),
    "lii_map_add_points": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32)]),
    "lii_map_delete_boxes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]),
    "lii_map_size": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "lii_map_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]),
    "lii_map_commit": (C.c_int, [C.c_void_p]),
    "lii_scan_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
    "lii_scan_upload_next": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
    "lii_scan_advance": (C.c_int, [C.c_void_p]),
    "lii_scan_set_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    "lii_undistort_imu": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32] + [C.c_void_p] * 4),
    "lii_undistort_cv": (C.c_int, [C.c_void_p] + [C.c_void_p] * 3),
    "lii_downsample": (C.c_int, [C.c_void_p, C.c_float, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "lii_downsample_skip": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "lii_scan_download": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]),
    "lii_ingest_pcl2": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(lii_pc2_fields), C.POINTER(lii_ingest_opts),
                                  C.POINTER(lii_frame_info), C.c_int32, C.POINTER(C.c_int32)]),
    "lii_ingest_livox": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(lii_livox_fields), C.POINTER(lii_ingest_opts),
                                   C.POINTER(lii_frame_info), C.c_int32, C.POINTER(C.c_int32)]),
    "lii_frame_select": (C.c_int, [C.c_void_p, C.c_int32]),
    "lii_ingest_pcl2_begin": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(lii_pc2_fields), C.POINTER(lii_ingest_opts)]),
    "lii_ingest_livox_begin": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(lii_livox_fields), C.POINTER(lii_ingest_opts)]),
    "lii_ingest_end": (C.c_int, [C.c_void_p, C.POINTER(lii_frame_info), C.c_int32, C.POINTER(C.c_int32)]),
    "lii_iekf_iterate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "lii_iekf_update": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(lii_iekf_opts),
                                  C.POINTER(lii_iekf_report)]),
    "lii_scan_register": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(lii_iekf_report)]),
    "lii_neighbors_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]),
    "lii_last_solve_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "lii_last_unfinished_queries": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "lii_map_incremental": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "lii_calib_set_buffers": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]),
    "lii_calib_eval": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]),
    "lii_calib_solve_stage": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(lii_calib_result)]),
    "lii_data_sufficiency": (C.c_int, [C.c_void_p, C.c_int32, C.c_double, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]),
    "lii_li_init_interpolate": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_double, C.c_void_p, C.c_void_p,
                                          C.POINTER(C.c_int32)]),
    "lii_li_init_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                  C.POINTER(lii_calib_result), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "lii_zero_phase_filter": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "lii_xcorr_lag": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]),
    "lii_li_init_set_device": (C.c_int, [C.c_void_p, C.c_int32]),
    "lii_comm_unique_id": (C.c_int, [C.c_void_p]),
    "lii_comm_init": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "lii_comm_init_ex": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32]),
    "lii_comm_describe": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int32]),
    "lii_comm_transport": (C.c_int, [C.c_void_p, C.c_void_p]),
    "lii_comm_rccl_ranks": (C.c_int, [C.c_void_p, C.c_void_p]),
    "lii_comm_set_partition": (C.c_int, [C.c_void_p, C.c_int32]),
    "lii_comm_destroy": (C.c_int, [C.c_void_p]),
    "lii_selftest_list_exchange": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.POINTER(C.c_int32), C.c_void_p, C.POINTER(C.c_int32), C.c_int32]),
    "lii_dev_alloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "lii_dev_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "lii_dev_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "lii_params_defaults": (C.c_int, [C.c_void_p]),
    "lii_params_load_yaml": (C.c_int, [C.c_char_p, C.c_void_p]),
    "lii_params_load_launch": (C.c_int, [C.c_char_p, C.c_char_p, C.c_void_p]),
    "lii_params_set": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p]),
    "lii_params_apply": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lii_params_last_error": (C.c_char_p, []),
    "lii_set_profiling": (C.c_int, [C.c_void_p, C.c_int32]),
    "lii_last_timings": (C.c_int, [C.c_void_p, C.c_void_p]),
    "lii_last_kernel_profile": (C.c_int, [C.c_void_p, C.c_void_p]),
}

EXPORTED_SYMBOLS = tuple(_DECLS)
_lib = None


def library_path() -> str:
    return _LIB


def load_library():
    """dlopen libliinit_hip.so and declare every C-ABI prototype.  Raises LIIError when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            raise LIIError(-2, f"{_LIB} not found — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(there is no Python/CPU fallback)")
        L = C.CDLL(_LIB)
        for name, (res, args) in _DECLS.items():
            if os.environ.get("LII_LIB") and not hasattr(L, name):
                continue  # (an older build under A/B measurement: what it lacks cannot be called)
            fn = getattr(L, name)  # AttributeError here = the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _ptr(a):
    # the raw address (ctypes converts an int for a c_void_p parameter); ndarray.ctypes.data_as() is two orders of
    # magnitude slower once a large framework is loaded in the process (measured: 60 us per call next to torch)
    return a.__array_interface__["data"][0]


def data_sufficiency(omg, data_accum_length):
    """LI_Init::data_sufficiency_assess (host function of the library): returns (eigenvalues, rot_percent, sufficient)."""
    w = np.ascontiguousarray(omg, np.float64).reshape(-1, 3)
    ev, rp, ok = np.zeros(3), np.zeros(3), C.c_int32(0)
    rc = load_library().lii_data_sufficiency(_ptr(w) if len(w) else None, len(w), float(data_accum_length), _ptr(ev), _ptr(rp), C.byref(ok))
    if rc != 0:
        raise LIIError(rc, "lii_data_sufficiency")
    return ev, rp, bool(ok.value)


class State:
    """StatesGroup (reference include/common_lib.h:68-169) as the 612-double `lii_state` POD."""

    def __init__(self, pod=None):
        if pod is None:
            pod = np.zeros(STATE_DOUBLES)
            pod[0:9] = np.eye(3).reshape(-1)
            pod[12:21] = np.eye(3).reshape(-1)
            cov = np.eye(24)
            cov[15:, 15:] = np.eye(9) * 0.00001  # INIT_COV, common_lib.h:79-80
            pod[36:] = cov.reshape(-1)
        self.pod = np.ascontiguousarray(pod, dtype=np.float64).copy()
        assert self.pod.shape == (STATE_DOUBLES,)

    def copy(self):
        return State(self.pod)

    rot_end = property(lambda s: s.pod[0:9].reshape(3, 3))
    pos_end = property(lambda s: s.pod[9:12])
    offset_R_L_I = property(lambda s: s.pod[12:21].reshape(3, 3))
    offset_T_L_I = property(lambda s: s.pod[21:24])
    vel_end = property(lambda s: s.pod[24:27])
    bias_g = property(lambda s: s.pod[27:30])
    bias_a = property(lambda s: s.pod[30:33])
    gravity = property(lambda s: s.pod[33:36])
    cov = property(lambda s: s.pod[36:].reshape(24, 24))


def pose6d_array(n):
    """n x 22 doubles: offset_time, acc[3], gyr[3], vel[3], pos[3], rot[9] (msg/Pose6D.msg)."""
    return np.zeros((n, 22))


def calib_state_array(n):
    """n x 22 doubles: rot_end[9], ang_vel[3], linear_vel[3], ang_acc[3], linear_acc[3], timestamp."""
    a = np.zeros((n, 22))
    a[:, 0] = a[:, 4] = a[:, 8] = 1.0
    return a


class Registrar:
    def __init__(self, max_scan_points=200_000, max_map_points=2_000_000, filter_size_map=0.15, map_cell_size=0.0,
                 device=0):
        self.L = load_library()
        n = C.c_int(0)
        rc = self.L.lii_device_count(C.byref(n))
        if rc != 0 or n.value <= 0:
            raise LIIError(-2, "no HIP device visible — libliinit_hip has no CPU fallback")
        cfg = lii_config()
        cfg.struct_size = C.sizeof(lii_config)
        cfg.device = device
        cfg.max_scan_points = int(max_scan_points)
        cfg.max_map_points = int(max_map_points)
        cfg.map_cell_size = float(map_cell_size)
        cfg.map_downsample_size = float(filter_size_map)
        cfg.max_match_dist2 = 5.0
        cfg.plane_threshold = 0.1
        cfg.laser_point_cov_inv = 1000.0
        self.h = C.c_void_p()
        self._check(self.L.lii_create(C.byref(cfg), C.byref(self.h)), None)
        self.max_scan_points = int(max_scan_points)
        self.max_map_points = int(max_map_points)
        self._dev_bufs = []

    # ------------------------------------------------------------------ plumbing
    def _check(self, rc, h=True):
        if rc != 0:
            msg = self.L.lii_last_error(self.h if h else None)
            raise LIIError(rc, (msg or b"").decode() or self.L.lii_strerror(rc).decode())

    def close(self):
        if getattr(self, "h", None):
            for p in self._dev_bufs:
                self.L.lii_dev_free(self.h, p)
            self._dev_bufs = []
            self.L.lii_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        self._check(self.L.lii_synchronize(self.h))

    # ------------------------------------------------------------------ local map
    def map_reset(self):
        self._check(self.L.lii_map_reset(self.h))

    def map_build(self, xyz):
        xyz = np.ascontiguousarray(xyz, np.float32)
        self._check(self.L.lii_map_build(self.h, _ptr(xyz), len(xyz), xyz.strides[0]))

    def map_add_points(self, xyz, downsample_on: bool) -> int:
        xyz = np.ascontiguousarray(xyz, np.float32)
        n = C.c_int32(0)
        self._check(self.L.lii_map_add_points(self.h, _ptr(xyz), len(xyz), xyz.strides[0] if len(xyz) else 12,
                                              int(downsample_on), C.byref(n)))
        return n.value

    def map_delete_boxes(self, boxes6) -> int:
        """ikdtree.Delete_Point_Boxes: boxes6 (n, 6) = min xyz, max xyz; returns the number of points removed."""
        b = np.ascontiguousarray(boxes6, np.float32).reshape(-1, 6)
        n = C.c_int32(0)
        self._check(self.L.lii_map_delete_boxes(self.h, _ptr(b) if len(b) else None, len(b), C.byref(n)))
        return n.value

    def map_size(self) -> int:
        n = C.c_int32(0)
        self._check(self.L.lii_map_size(self.h, C.byref(n)))
        return n.value

    def map_download(self):
        n = C.c_int32(0)
        self._check(self.L.lii_map_download(self.h, None, 0, C.byref(n)))
        out = np.zeros((max(n.value, 1), 3), np.float32)
        self._check(self.L.lii_map_download(self.h, _ptr(out), len(out), C.byref(n)))
        return out[:n.value]

    def map_commit(self):
        self._check(self.L.lii_map_commit(self.h))

    # ------------------------------------------------------------------ scan
    def scan_upload(self, pts):
        """pts: (n,4) float32 (x,y,z,t_ms) or a structured/2-D array with the 48-byte PointXYZINormal layout (n,12)."""
        pts = np.ascontiguousarray(pts, np.float32)
        assert pts.ndim == 2 and pts.shape[1] in (4, 12)
        toff = 12 if pts.shape[1] == 4 else 36
        self._check(self.L.lii_scan_upload(self.h, _ptr(pts), len(pts), pts.strides[0] if len(pts) else 16, toff))

    def scan_upload_next(self, pts):
        """Starts the NEXT scan on its way (copy stream, second device buffer); `pts` must stay alive and untouched until
        scan_advance() has returned (a pinned (n,4) source is read by the copy engine directly)."""
        assert pts.dtype == np.float32 and pts.flags.c_contiguous and pts.ndim == 2 and pts.shape[1] in (4, 12)
        toff = 12 if pts.shape[1] == 4 else 36
        self._check(self.L.lii_scan_upload_next(self.h, _ptr(pts), len(pts), pts.strides[0] if len(pts) else 16, toff))

    def scan_advance(self):
        """The scan handed to scan_upload_next becomes the current one (as scan_upload would have made it)."""
        self._check(self.L.lii_scan_advance(self.h))

    def device_scan(self, pts4):
        """Copies a float4 scan into a caller-owned device buffer; returns an opaque (ptr, n) for scan_set_device."""
        pts4 = np.ascontiguousarray(pts4, np.float32)
        assert pts4.ndim == 2 and pts4.shape[1] == 4
        p = C.c_void_p()
        self._check(self.L.lii_dev_alloc(self.h, max(pts4.nbytes, 16), C.byref(p)))
        self._dev_bufs.append(p)
        if len(pts4):
            self._check(self.L.lii_dev_upload(self.h, p, _ptr(pts4), pts4.nbytes))
        return (p, len(pts4))

    def scan_set_device(self, dev):
        self._check(self.L.lii_scan_set_device(self.h, dev[0], dev[1]))

    def undistort_imu(self, poses22, end_R, end_p, R_LI, T_LI):
        poses = np.ascontiguousarray(poses22, np.float64).reshape(-1, 22)
        a = [np.ascontiguousarray(x, np.float64).reshape(-1) for x in (end_R, end_p, R_LI, T_LI)]
        self._check(self.L.lii_undistort_imu(self.h, _ptr(poses), len(poses), *[_ptr(x) for x in a]))

    def undistort_cv(self, omega, vel, end_R):
        a = [np.ascontiguousarray(x, np.float64).reshape(-1) for x in (omega, vel, end_R)]
        self._check(self.L.lii_undistort_cv(self.h, *[_ptr(x) for x in a]))

    def downsample(self, leaf: float, want_count: bool = True):
        """Voxel-grid filter.  want_count=False keeps it asynchronous: the size of the result stays on the device
        (the registration kernels read it there) and no host synchronisation happens."""
        if not want_count:
            self._check(self.L.lii_downsample(self.h, float(leaf), None, None))
            return None
        n, f = C.c_int32(0), C.c_int32(0)
        self._check(self.L.lii_downsample(self.h, float(leaf), C.byref(n), C.byref(f)))
        return n.value, bool(f.value)

    def downsample_skip(self) -> int:
        n = C.c_int32(0)
        self._check(self.L.lii_downsample_skip(self.h, C.byref(n)))
        return n.value

    def scan_download(self, which=0):
        n = C.c_int32(0)
        self._check(self.L.lii_scan_download(self.h, which, None, 0, C.byref(n)))
        out = np.zeros((max(n.value, 1), 4), np.float32)
        self._check(self.L.lii_scan_download(self.h, which, _ptr(out), len(out), C.byref(n)))
        return out[:n.value]

    # ------------------------------------------------------------------ ingest
    def _ingest(self, fn, data, n_points, fields, lidar_type, n_scans, point_filter_num, blind, stamp_s, cut_frame_num,
                scan_count):
        raw = np.frombuffer(data, np.uint8)
        opts = lii_ingest_opts(C.sizeof(lii_ingest_opts), lidar_type, n_scans, point_filter_num, blind, stamp_s,
                               cut_frame_num, scan_count)
        frames = (lii_frame_info * 64)()
        nf = C.c_int32(0)
        self._check(fn(self.h, _ptr(raw) if len(raw) else None, n_points, C.byref(fields), C.byref(opts), frames, 64, C.byref(nf)))
        self.frame_tail_ms = [frames[k].last_offset_ms for k in range(nf.value)]  # curvature of each frame's last point
        return [(frames[k].begin_time_s, frames[k].offset, frames[k].count) for k in range(nf.value)]

    def ingest_pcl2(self, data, n_points, fields, lidar_type, n_scans, point_filter_num, blind, stamp_s, cut_frame_num=1,
                    scan_count=1000):
        """process_cut_frame_pcl2 on the device.  Returns [(begin_time_s, offset, count)] per sub-frame."""
        return self._ingest(self.L.lii_ingest_pcl2, data, n_points, lii_pc2_fields(*fields), lidar_type, n_scans,
                            point_filter_num, blind, stamp_s, cut_frame_num, scan_count)

    def ingest_livox(self, data, n_points, fields, n_scans, point_filter_num, blind, stamp_s, cut_frame_num=1, scan_count=1000):
        """process_cut_frame_livox on the device."""
        return self._ingest(self.L.lii_ingest_livox, data, n_points, lii_livox_fields(*fields), 1, n_scans, point_filter_num,
                            blind, stamp_s, cut_frame_num, scan_count)

    def frame_select(self, k: int):
        self._check(self.L.lii_frame_select(self.h, k))

    def ingest_pcl2_begin(self, data, n_points, fields, lidar_type, n_scans, point_filter_num, blind, stamp_s, cut_frame_num=1,
                          scan_count=1000):
        """The overlapped form (ABI 9): the message is put under way and the call returns; `data` (a buffer object) is kept alive here
        until the matching ingest_end."""
        raw = np.frombuffer(data, np.uint8)
        opts = lii_ingest_opts(C.sizeof(lii_ingest_opts), lidar_type, n_scans, point_filter_num, blind, stamp_s, cut_frame_num, scan_count)
        f = lii_pc2_fields(*fields)
        self._check(self.L.lii_ingest_pcl2_begin(self.h, _ptr(raw) if len(raw) else None, n_points, C.byref(f), C.byref(opts)))
        self._ingest_keep = getattr(self, "_ingest_keep", []) + [raw]

    def ingest_livox_begin(self, data, n_points, fields, n_scans, point_filter_num, blind, stamp_s, cut_frame_num=1, scan_count=1000):
        raw = np.frombuffer(data, np.uint8)
        opts = lii_ingest_opts(C.sizeof(lii_ingest_opts), 1, n_scans, point_filter_num, blind, stamp_s, cut_frame_num, scan_count)
        f = lii_livox_fields(*fields)
        self._check(self.L.lii_ingest_livox_begin(self.h, _ptr(raw) if len(raw) else None, n_points, C.byref(f), C.byref(opts)))
        self._ingest_keep = getattr(self, "_ingest_keep", []) + [raw]

    def ingest_end(self):
        """Waits for the oldest message under way; its frames become the ones frame_select serves.  Returns what ingest_pcl2 returns."""
        frames = (lii_frame_info * 64)()
        nf = C.c_int32(0)
        self._check(self.L.lii_ingest_end(self.h, frames, 64, C.byref(nf)))
        if getattr(self, "_ingest_keep", None):
            self._ingest_keep.pop(0)
        self.frame_tail_ms = [frames[k].last_offset_ms for k in range(nf.value)]
        return [(frames[k].begin_time_s, frames[k].offset, frames[k].count) for k in range(nf.value)]

    # ------------------------------------------------------------------ registration
    def iekf_iterate(self, state: State, search: bool, imu_en: bool):
        out = np.zeros(91)
        self._check(self.L.lii_iekf_iterate(self.h, _ptr(state.pod), int(search), int(imu_en), _ptr(out)))
        return out

    def iekf_update(self, state: State, state_prop: State, max_iterations=4, imu_en=False):
        """Runs the whole iterated update in place on `state`; returns the report dict."""
        opts = lii_iekf_opts(int(max_iterations), int(imu_en))
        rep = lii_iekf_report()
        self._check(self.L.lii_iekf_update(self.h, _ptr(state.pod), _ptr(state_prop.pod), C.byref(opts), C.byref(rep)))
        return dict(iterations=rep.iterations, searches=rep.searches, effect_num=rep.effect_num,
                    converged=bool(rep.converged), normal_eq=np.array(rep.normal_eq[:]))

    def scan_register(self, state: State, state_prop: State, *, imu_poses=None, cv=False, leaf=0.0, max_iterations=4,
                      imu_en=False, scan_dev=None, scan_sorted=False, map_update=False, next_scan=None, while_waiting=None):
        """Undistortion + voxel grid + iterated update in one library call (one host synchronisation).  scan_dev: a
        device_scan() handle to adopt first (what scan_set_device would do, without the separate call).  scan_sorted: the
        points are in ascending time order (lii_scan_job::scan_sorted).  map_update: map_incremental with the final state
        follows inside the call (lii_scan_job::map_update) - do not call map_incremental() for this scan."""
        job = lii_scan_job()
        job.struct_size = C.sizeof(lii_scan_job)
        job.scan_sorted = 1 if scan_sorted else 0
        job.map_update = 1 if map_update else 0
        if scan_dev is not None:
            job.scan_dev, job.n_scan_dev = scan_dev[0], scan_dev[1]
        if next_scan is not None:  # a device_scan() handle: the scan the NEXT call will bring (lii_scan_job::next_scan_dev - its prologue is pre-armed)
            job.next_scan_dev, job.next_n_scan = next_scan[0], next_scan[1]
        hook = None
        if while_waiting is not None:  # a Python callable: runs once inside the call, when every launch is enqueued (lii_scan_job::while_waiting)
            hook = WAIT_HOOK(lambda _arg: while_waiting())
            job.while_waiting = C.cast(hook, C.c_void_p)
        poses = None
        if imu_poses is not None:
            poses = np.ascontiguousarray(imu_poses, np.float64).reshape(-1, 22)
            job.undistort, job.imu_poses, job.n_imu_poses = 1, _ptr(poses), len(poses)
        elif cv:
            job.undistort = 2
        job.leaf = float(leaf)
        job.opts = lii_iekf_opts(int(max_iterations), int(imu_en))
        rep = lii_iekf_report()
        self._check(self.L.lii_scan_register(self.h, C.byref(job), _ptr(state.pod), _ptr(state_prop.pod), C.byref(rep)))
        return dict(iterations=rep.iterations, searches=rep.searches, effect_num=rep.effect_num,
                    converged=bool(rep.converged), normal_eq=np.array(rep.normal_eq[:]))

    def neighbors(self, n):
        pts = np.zeros((n, 5, 3), np.float32)
        cnt = np.zeros(n, np.int32)
        sel = np.zeros(n, np.uint8)
        self._check(self.L.lii_neighbors_download(self.h, _ptr(pts), _ptr(cnt), _ptr(sel), n))
        return pts, cnt, sel

    def map_incremental(self, state: State, want_counts: bool = True):
        """want_counts=False: both size pointers NULL - the update is enqueued without a host round trip (predicted list sizes)."""
        if not want_counts:
            self._check(self.L.lii_map_incremental(self.h, _ptr(state.pod), None, None))
            return None
        a, b = C.c_int32(0), C.c_int32(0)
        self._check(self.L.lii_map_incremental(self.h, _ptr(state.pod), C.byref(a), C.byref(b)))
        return a.value, b.value

    # ------------------------------------------------------------------ calibration
    def calib_set_buffers(self, imu22, lidar22):
        imu = np.ascontiguousarray(imu22, np.float64).reshape(-1, 22)
        lid = np.ascontiguousarray(lidar22, np.float64).reshape(-1, 22)
        assert len(imu) == len(lid)
        self._check(self.L.lii_calib_set_buffers(self.h, _ptr(imu), _ptr(lid), len(imu)))

    def calib_eval(self, stage, params):
        dof = {1: 3, 2: 7, 3: 9}[stage]
        p = np.ascontiguousarray(params, np.float64).reshape(-1)
        JtJ, Jtr, cost = np.zeros((dof, dof)), np.zeros(dof), C.c_double(0)
        self._check(self.L.lii_calib_eval(self.h, stage, _ptr(p), _ptr(JtJ), _ptr(Jtr), C.byref(cost)))
        return JtJ, Jtr, cost.value

    def calib_solve_stage(self, stage, result=None):
        res = result if result is not None else lii_calib_result()
        if result is None:
            res.R_LI[:] = list(np.eye(3).reshape(-1))
        self._check(self.L.lii_calib_solve_stage(self.h, stage, C.byref(res)))
        return res

    def li_init_run(self, imu22, lidar22, orig_odom_freq, cut_frame_num):
        """LI_Init::LI_Initialization after downsample_interpolate_IMU; returns (lii_calib_result, time_lag_1, total_lag)."""
        imu = np.ascontiguousarray(imu22, np.float64).reshape(-1, 22)
        lid = np.ascontiguousarray(lidar22, np.float64).reshape(-1, 22)
        res = lii_calib_result()
        l1, tot = C.c_double(0), C.c_double(0)
        self._check(self.L.lii_li_init_run(self.h, _ptr(imu), _ptr(lid), len(imu), int(orig_odom_freq), int(cut_frame_num),
                                           C.byref(res), C.byref(l1), C.byref(tot)))
        return res, l1.value, tot.value

    def zero_phase_filter(self, seqs22):
        """LI_Init::zero_phase_filt on the device: seqs22 (n_seq, n, 22) -> filtered copy."""
        a = np.ascontiguousarray(seqs22, np.float64)
        assert a.ndim == 3 and a.shape[2] == 22
        out = np.zeros_like(a)
        self._check(self.L.lii_zero_phase_filter(self.h, _ptr(a), a.shape[0], a.shape[1], _ptr(out)))
        return out

    def xcorr_lag(self, imu22, lidar22) -> int:
        """LI_Init::xcorr_temporal_init on the device: lag_IMU_wtr_Lidar in samples."""
        imu = np.ascontiguousarray(imu22, np.float64).reshape(-1, 22)
        lid = np.ascontiguousarray(lidar22, np.float64).reshape(-1, 22)
        lag = C.c_int32(0)
        self._check(self.L.lii_xcorr_lag(self.h, _ptr(imu), _ptr(lid), len(imu), C.byref(lag)))
        return lag.value

    def li_init_set_device(self, on: bool):
        self._check(self.L.lii_li_init_set_device(self.h, int(on)))

    # ------------------------------------------------------------------ multi-GPU
    def comm_unique_id(self) -> bytes:
        buf = (C.c_uint8 * 128)()
        self._check(self.L.lii_comm_unique_id(buf), None)
        return bytes(buf)

    def comm_init(self, n_ranks, rank, uid: bytes, transport="auto"):
        """transport: "auto" (peer-mapped HBM mailbox when all ranks share the node, else the host-memory mailbox, else RCCL),
        "rccl", "mailbox" (HBM, over HIP IPC), "mailbox_host"."""
        buf = (C.c_uint8 * 128).from_buffer_copy(uid)
        self._check(self.L.lii_comm_init_ex(self.h, n_ranks, rank, buf, {"auto": 0, "rccl": 1, "mailbox": 2, "mailbox_host": 3}[transport]))

    def comm_set_partition(self, library_partition):
        """True / 1 / "index" (default): every rank hands over the whole scan, the library splits the down-sampled cloud into
        contiguous blocks; 2 / "voxel": ... by voxel - a rank filters and registers the voxels whose key hashes to it (where the
        filter is fused into the de-skew: lii_scan_register; by index elsewhere); False / 0 / "caller": the caller hands every
        rank its own points."""
        mode = {"caller": 0, "index": 1, "voxel": 2}.get(library_partition, library_partition)
        self._check(self.L.lii_comm_set_partition(self.h, int(mode)))

    def comm_transport(self) -> str:
        t = C.c_int32(0)
        self._check(self.L.lii_comm_transport(self.h, C.byref(t)))
        return {0: "none", 1: "rccl", 2: "mailbox", 3: "mailbox_host"}[t.value]

    def comm_describe(self) -> str:
        buf = C.create_string_buffer(512)
        self._check(self.L.lii_comm_describe(self.h, buf, 512))
        return buf.value.decode()

    def comm_rccl_ranks(self) -> int:
        """ncclCommCount of the attached RCCL communicator (0: no RCCL communicator)."""
        n = C.c_int32(0)
        self._check(self.L.lii_comm_rccl_ranks(self.h, C.byref(n)))
        return int(n.value)

    def last_solve_info(self):
        """Passes of the last device-resident update whose 12 x 12 elimination needed the routine with row exchanges."""
        n = C.c_int32(0)
        self._check(self.L.lii_last_solve_info(self.h, C.byref(n)))
        return int(n.value)

    def last_unfinished_queries(self):
        """Queries the most recent search pass left to the fit launch behind it."""
        n = C.c_int32(0)
        self._check(self.L.lii_last_unfinished_queries(self.h, C.byref(n)))
        return int(n.value)

    def selftest_list_exchange(self, add_lists, nodown_lists, form):
        """The list exchange of a sharded job's map update played by ONE handle for len(add_lists) ranks (form "gather": the mailbox
        transport's gather areas, "allgather": the trimmed all-gather layout of the RCCL transport); rank r's lists are (n, 4) float32.
        Returns the joined (add, nodown) lists every played rank ended with."""
        n = len(add_lists)
        assert n == len(nodown_lists) and n >= 1
        na = np.array([len(a) for a in add_lists], np.int32)
        nn = np.array([len(a) for a in nodown_lists], np.int32)
        cat = lambda ls: np.ascontiguousarray(np.concatenate([np.asarray(a, np.float32).reshape(-1, 4) for a in ls] + [np.zeros((1, 4), np.float32)]))
        a, d = cat(add_lists), cat(nodown_lists)
        cap = int(max(na.sum(), nn.sum(), 1))
        oa, od = np.zeros((cap, 4), np.float32), np.zeros((cap, 4), np.float32)
        ona, onn = C.c_int32(0), C.c_int32(0)
        self._check(self.L.lii_selftest_list_exchange(self.h, n, {"gather": 0, "allgather": 1}[form], _ptr(a), _ptr(na), _ptr(d), _ptr(nn),
                                                      _ptr(oa), C.byref(ona), _ptr(od), C.byref(onn), cap))
        return oa[:ona.value].copy(), od[:onn.value].copy()

    def comm_destroy(self):
        self._check(self.L.lii_comm_destroy(self.h))

    def set_profiling(self, enabled: bool):
        """Turns the HIP-event timing of the registration kernels on/off and zeroes the accumulators."""
        self._check(self.L.lii_set_profiling(self.h, int(enabled)))  # 1 = start (zero), 2 = resume, 0 = pause

    def timings(self):
        """[0] sum ms search-pass kernel, [1] sum ms residual-pass kernel, [2] sum ms reduce kernel, [3] host solve ms of
        the last update, [4] total ms of the last update, [5]/[6] launch counts behind [0]/[1]."""
        out = np.zeros(8)
        self._check(self.L.lii_last_timings(self.h, _ptr(out)))
        return out

    def kernel_profile(self):
        """Per-launch brackets of lii_scan_register collected under set_profiling(3): {kind: (total ms, launches)}, scans."""
        kp = lii_kernel_profile()
        kp.struct_size = C.sizeof(lii_kernel_profile)
        self._check(self.L.lii_last_kernel_profile(self.h, C.byref(kp)))
        return {k: (kp.ms[i], kp.launches[i]) for i, k in enumerate(KERNEL_KINDS)}, kp.scans
