"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes front-end of the CPU restatement (oracle/_build/liboracle.so) and, when it has been built,
of the unmodified reference ikd-Tree (oracle/_ref/libref_ikdtree.so).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product package
`lidar_imu_init_amd` never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")
_REF_PATH = os.path.join(_HERE, "_ref", "libref_ikdtree.so")

STATE_DOUBLES = 36 + 24 * 24
LOG_DOUBLES = 116


def build(force: bool = False) -> None:
    """Compile the oracle (and oracle/_ref when /root/reference is present)."""
    if force or not os.path.exists(_LIB_PATH) or _stale():
        subprocess.check_call(["make", "-C", _HERE, "oracle"], stdout=subprocess.DEVNULL)
    if os.path.exists("/root/reference/include/ikd-Tree/ikd_Tree.cpp") and (force or not os.path.exists(_REF_PATH)):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


def _stale() -> bool:
    t = os.path.getmtime(_LIB_PATH)
    for f in os.listdir(_HERE):
        if f.endswith((".hpp", ".cpp")) and not f.startswith("ref_") and os.path.getmtime(os.path.join(_HERE, f)) > t:
            return True
    return False


_lib = None
_ref = None


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def _up(a):
    return a.ctypes.data_as(C.POINTER(C.c_ubyte))


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.orc_tree_create.restype = C.c_void_p
        for name in ("orc_tree_destroy", "orc_tree_set_downsample", "orc_tree_build", "orc_tree_add_points",
                     "orc_tree_size", "orc_tree_validnum", "orc_tree_flatten", "orc_tree_knn", "orc_iekf_update",
                     "orc_iekf_iterate_once", "orc_map_incremental"):
            getattr(L, name).argtypes = None
        L.orc_tree_destroy.argtypes = [C.c_void_p]
        L.orc_tree_set_downsample.argtypes = [C.c_void_p, C.c_float]
        L.orc_tree_build.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int]
        L.orc_tree_add_points.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int, C.c_int]
        L.orc_tree_size.argtypes = [C.c_void_p]
        L.orc_tree_validnum.argtypes = [C.c_void_p]
        L.orc_tree_flatten.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int]
        L.orc_tree_knn.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int, C.c_int, C.c_double,
                                   C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int), C.c_int]
        L.orc_exp.argtypes = [C.POINTER(C.c_double), C.c_double, C.POINTER(C.c_double)]
        L.orc_exp1.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_exp3.argtypes = [C.c_double, C.c_double, C.c_double, C.POINTER(C.c_double)]
        L.orc_log.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_rot_to_euler.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_state_init.argtypes = [C.POINTER(C.c_double)]
        L.orc_state_boxplus.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_state_boxminus.argtypes = [C.POINTER(C.c_double)] * 3
        L.orc_inverse.argtypes = [C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double)]
        L.orc_esti_plane.argtypes = [C.POINTER(C.c_float), C.c_double, C.POINTER(C.c_double)]
        L.orc_esti_plane_batch.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_double, C.POINTER(C.c_double),
                                           C.POINTER(C.c_ubyte)]
        L.orc_sort_by_time.argtypes = [C.POINTER(C.c_float), C.c_int]
        L.orc_ingest_pcl2.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                                      C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_ingest_livox.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double,
                                       C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_undistort_imu.argtypes = [C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_double), C.c_int] + \
                                       [C.POINTER(C.c_double)] * 4
        L.orc_undistort_cv.argtypes = [C.POINTER(C.c_float), C.c_int] + [C.POINTER(C.c_double)] * 3
        L.orc_voxel_grid.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_float, C.POINTER(C.c_float),
                                     C.POINTER(C.c_int)]
        L.orc_iekf_update.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_double),
                                      C.POINTER(C.c_double), C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_int),
                                      C.POINTER(C.c_ubyte), C.POINTER(C.c_float), C.POINTER(C.c_float),
                                      C.POINTER(C.c_double)]
        L.orc_iekf_iterate_once.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_double),
                                            C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_float),
                                            C.POINTER(C.c_int), C.POINTER(C.c_ubyte), C.POINTER(C.c_float),
                                            C.POINTER(C.c_double)]
        L.orc_map_incremental.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_double),
                                          C.c_float, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_int),
                                          C.POINTER(C.c_float), C.POINTER(C.c_int)]
        _lib = L
    return _lib


def ref_available() -> bool:
    return os.path.exists(_REF_PATH)


def ref():
    """The unmodified reference ikd-Tree (None when oracle/_ref has not been built)."""
    global _ref
    if _ref is None:
        if not ref_available():
            build()
        if not ref_available():
            return None
        R = C.CDLL(_REF_PATH)
        R.ref_tree_create.restype = C.c_void_p
        R.ref_tree_destroy.argtypes = [C.c_void_p]
        R.ref_tree_set_downsample.argtypes = [C.c_void_p, C.c_float]
        R.ref_tree_build.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int]
        R.ref_tree_add_points.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int, C.c_int]
        R.ref_tree_size.argtypes = [C.c_void_p]
        R.ref_tree_delete_boxes.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int]
        R.ref_tree_validnum.argtypes = [C.c_void_p]
        R.ref_tree_flatten.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int, C.c_int]
        R.ref_tree_knn.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int, C.c_int, C.c_double,
                                   C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int), C.c_int]
        _ref = R
    return _ref


def _f32(a, cols=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if cols is not None:
        assert a.ndim == 2 and a.shape[1] == cols, a.shape
    return a


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


# ------------------------------------------------------------------------------------------------
class Tree:
    """Restated incremental k-d tree (or, with backend='ref', the reference's own ikd-Tree)."""

    def __init__(self, backend: str = "oracle", downsample: float | None = None):
        self.backend = backend
        if backend == "oracle":
            self._L = lib()
            self._p = "orc_tree_"
        elif backend == "ref":
            self._L = ref()
            if self._L is None:
                raise RuntimeError("oracle/_ref/libref_ikdtree.so is not built")
            self._p = "ref_tree_"
        else:
            raise ValueError(backend)
        self._h = C.c_void_p(getattr(self._L, self._p + "create")())
        if downsample is not None:
            self.set_downsample(downsample)

    def _f(self, name):
        return getattr(self._L, self._p + name)

    def close(self):
        if self._h:
            self._f("destroy")(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_downsample(self, box: float):
        self._f("set_downsample")(self._h, C.c_float(box))

    def build(self, xyz):
        xyz = _f32(xyz, 3)
        self._f("build")(self._h, _fp(xyz), len(xyz))

    def add_points(self, xyz, downsample_on: bool) -> int:
        xyz = _f32(xyz, 3)
        if len(xyz) == 0:
            return 0
        return self._f("add_points")(self._h, _fp(xyz), len(xyz), int(downsample_on))

    def delete_boxes(self, boxes6) -> int:
        """KD_TREE::Delete_Point_Boxes of the unmodified reference tree (backend 'ref' only)."""
        assert self.backend == "ref"
        b = _f32(np.asarray(boxes6).reshape(-1, 6))
        return self._L.ref_tree_delete_boxes(self._h, _fp(b), len(b))

    def size(self) -> int:
        return self._f("size")(self._h)

    def validnum(self) -> int:
        return self._f("validnum")(self._h)

    def flatten(self, settle_ms: int = 200):
        cap = max(self.size(), 1)
        out = np.zeros((cap, 3), np.float32)
        if self.backend == "ref":
            n = self._f("flatten")(self._h, _fp(out), cap, settle_ms)
        else:
            n = self._f("flatten")(self._h, _fp(out), cap)
        return out[:n].copy()

    def knn(self, q, k: int = 5, max_dist: float = 5.0, threads: int = 1):
        q = _f32(q, 3)
        n = len(q)
        pts = np.zeros((n, k, 3), np.float32)
        d2 = np.zeros((n, k), np.float32)
        cnt = np.zeros(n, np.int32)
        self._f("knn")(self._h, _fp(q), n, k, C.c_double(max_dist), _fp(pts), _fp(d2), _ip(cnt), threads)
        return pts, d2, cnt

    # --- oracle-only -------------------------------------------------------------------------
    def iekf_update(self, body4, state, state_prop, max_iterations=4, imu_en=False, threads=1, literal_gain=False):
        assert self.backend == "oracle"
        body4 = _f32(body4, 4)
        n = len(body4)
        st = _f64(state).copy()
        sp = _f64(state_prop)
        logs = np.zeros((max_iterations, LOG_DOUBLES))
        nearest = np.zeros((n, 5, 3), np.float32)
        nn = np.zeros(n, np.int32)
        sel = np.zeros(n, np.uint8)
        normvec = np.zeros((n, 4), np.float32)
        world = np.zeros((n, 3), np.float32)
        sec = C.c_double(0)
        it = self._L.orc_iekf_update(self._h, _fp(body4), n, _dp(st), _dp(sp), max_iterations, int(imu_en), threads,
                                     int(literal_gain), _dp(logs), max_iterations, _fp(nearest), _ip(nn), _up(sel),
                                     _fp(normvec), _fp(world), C.byref(sec))
        return dict(state=st, iters=it, logs=logs[:it], nearest=nearest, nearest_n=nn, selected=sel,
                    normvec=normvec, world=world, seconds=sec.value)

    def iterate_once(self, body4, state, search=True, imu_en=False, threads=1, selected=None):
        assert self.backend == "oracle"
        body4 = _f32(body4, 4)
        n = len(body4)
        out91 = np.zeros(91)
        nearest = np.zeros((n, 5, 3), np.float32)
        nn = np.zeros(n, np.int32)
        sel = np.ones(n, np.uint8) if selected is None else np.ascontiguousarray(selected, np.uint8).copy()
        normvec = np.zeros((n, 4), np.float32)
        pabcd = np.zeros((n, 4))
        self._L.orc_iekf_iterate_once(self._h, _fp(body4), n, _dp(_f64(state)), int(search), int(imu_en), threads,
                                      _dp(out91), _fp(nearest), _ip(nn), _up(sel), _fp(normvec), _dp(pabcd))
        return dict(out91=out91, nearest=nearest, nearest_n=nn, selected=sel, normvec=normvec, pabcd=pabcd)

    def map_incremental(self, body4, state, filter_size_map: float, apply: bool = True):
        assert self.backend == "oracle"
        body4 = _f32(body4, 4)
        n = len(body4)
        a = np.zeros((n, 3), np.float32)
        b = np.zeros((n, 3), np.float32)
        na, nb = C.c_int(0), C.c_int(0)
        self._L.orc_map_incremental(self._h, _fp(body4), n, _dp(_f64(state)), C.c_float(filter_size_map), int(apply),
                                    _fp(a), C.byref(na), _fp(b), C.byref(nb))
        return a[:na.value].copy(), b[:nb.value].copy()


# ------------------------------------------------------------------------------------------------
def state_init():
    s = np.zeros(STATE_DOUBLES)
    lib().orc_state_init(_dp(s))
    return s


class StateView:
    """Named views into the 612-double StatesGroup POD (reference include/common_lib.h:68-169)."""

    def __init__(self, pod):
        self.pod = pod

    rot_end = property(lambda s: s.pod[0:9].reshape(3, 3))
    pos_end = property(lambda s: s.pod[9:12])
    offset_R_L_I = property(lambda s: s.pod[12:21].reshape(3, 3))
    offset_T_L_I = property(lambda s: s.pod[21:24])
    vel_end = property(lambda s: s.pod[24:27])
    bias_g = property(lambda s: s.pod[27:30])
    bias_a = property(lambda s: s.pod[30:33])
    gravity = property(lambda s: s.pod[33:36])
    cov = property(lambda s: s.pod[36:].reshape(24, 24))


def state_boxplus(state, d24):
    s = _f64(state).copy()
    lib().orc_state_boxplus(_dp(s), _dp(_f64(d24)))
    return s


def state_boxminus(a, b):
    out = np.zeros(24)
    lib().orc_state_boxminus(_dp(_f64(a)), _dp(_f64(b)), _dp(out))
    return out


def exp_so3(w, dt=None):
    R = np.zeros(9)
    w = _f64(w)
    if dt is None:
        lib().orc_exp1(_dp(w), _dp(R))
    else:
        lib().orc_exp(_dp(w), C.c_double(dt), _dp(R))
    return R.reshape(3, 3)


def exp3(a, b, c):
    R = np.zeros(9)
    lib().orc_exp3(a, b, c, _dp(R))
    return R.reshape(3, 3)


def log_so3(R):
    out = np.zeros(3)
    lib().orc_log(_dp(_f64(R).reshape(-1)), _dp(out))
    return out


def rot_to_euler(R):
    out = np.zeros(3)
    lib().orc_rot_to_euler(_dp(_f64(R).reshape(-1)), _dp(out))
    return out


def inverse(A):
    A = _f64(A)
    out = np.zeros_like(A)
    lib().orc_inverse(_dp(A), A.shape[0], _dp(out))
    return out


def esti_plane(pts5x3, threshold=0.1):
    p = _f32(pts5x3).reshape(-1)
    out = np.zeros(4)
    ok = lib().orc_esti_plane(_fp(p), C.c_double(threshold), _dp(out))
    return bool(ok), out


def esti_plane_batch(pts_nx5x3, threshold=0.1):
    p = _f32(pts_nx5x3).reshape(-1, 15)
    n = len(p)
    out = np.zeros((n, 4))
    valid = np.zeros(n, np.uint8)
    lib().orc_esti_plane_batch(_fp(p), n, C.c_double(threshold), _dp(out), _up(valid))
    return valid.astype(bool), out


def sort_by_time(pts4):
    p = _f32(pts4, 4).copy()
    lib().orc_sort_by_time(_fp(p), len(p))
    return p


def undistort_imu(pts4, poses22, end_R, end_p, R_LI, T_LI):
    """Time-sorts (stable) then back-propagates; returns the sorted, de-skewed float4 cloud."""
    p = _f32(pts4, 4).copy()
    poses = _f64(poses22).reshape(-1, 22)
    lib().orc_undistort_imu(_fp(p), len(p), _dp(poses), len(poses), _dp(_f64(end_R).reshape(-1)), _dp(_f64(end_p)),
                            _dp(_f64(R_LI).reshape(-1)), _dp(_f64(T_LI)))
    return p


def undistort_cv(pts4, omega, vel, end_R):
    p = _f32(pts4, 4).copy()
    lib().orc_undistort_cv(_fp(p), len(p), _dp(_f64(omega)), _dp(_f64(vel)), _dp(_f64(end_R).reshape(-1)))
    return p


def voxel_grid(pts4, leaf):
    p = _f32(pts4, 4)
    out = np.zeros_like(p)
    n = C.c_int(0)
    filtered = lib().orc_voxel_grid(_fp(p), len(p), C.c_float(leaf), _fp(out), C.byref(n))
    return out[:n.value].copy(), bool(filtered)


# LID_TYPE, reference include/common_lib.h:55
AVIA, VELO, OUSTER, L515, PANDAR, ROBOSENSE = 1, 2, 3, 4, 5, 6


def _unflatten(nf, out, begin, offs, cnts):
    if nf < 0:
        raise RuntimeError(f"oracle ingest failed ({nf})")
    return [(begin[k], out[offs[k]:offs[k] + cnts[k]].copy()) for k in range(nf)]


def ingest_pcl2(data, n_points, fields, lidar_type, n_scans, point_filter_num, blind, stamp_s, cut_frame_num, scan_count):
    """process_cut_frame_pcl2 (reference src/preprocess.cpp:115-335).  data: the raw bytes of PointCloud2::data;
    fields = (point_step, off_x, off_y, off_z, off_intensity, off_time, off_ring).
    Returns [(begin_time_ms, float32 (m,4) array of x,y,z,curvature_ms), ...] — one entry per sub-frame."""
    raw = np.ascontiguousarray(np.frombuffer(data, np.uint8))
    f = np.asarray(fields, np.int32)
    out = np.zeros((max(n_points, 1), 4), np.float32)
    cap_f = max(int(cut_frame_num), 1) + 1
    begin, offs, cnts = np.zeros(cap_f), np.zeros(cap_f, np.int32), np.zeros(cap_f, np.int32)
    nf = lib().orc_ingest_pcl2(raw.ctypes.data, n_points, f.ctypes.data, lidar_type, n_scans, point_filter_num, blind,
                               stamp_s, cut_frame_num, scan_count, out.ctypes.data, len(out), begin.ctypes.data,
                               offs.ctypes.data, cnts.ctypes.data, cap_f)
    return _unflatten(nf, out, begin, offs, cnts)


def ingest_livox(data, n_points, fields, n_scans, point_filter_num, blind, stamp_s, cut_frame_num, scan_count):
    """process_cut_frame_livox (reference src/preprocess.cpp:50-113).
    fields = (point_step, off_offset_time, off_x, off_y, off_z, off_reflectivity, off_tag, off_line)."""
    raw = np.ascontiguousarray(np.frombuffer(data, np.uint8))
    f = np.asarray(fields, np.int32)
    out = np.zeros((max(n_points, 1), 4), np.float32)
    cap_f = max(int(cut_frame_num), 1) + 1
    begin, offs, cnts = np.zeros(cap_f), np.zeros(cap_f, np.int32), np.zeros(cap_f, np.int32)
    nf = lib().orc_ingest_livox(raw.ctypes.data, n_points, f.ctypes.data, n_scans, point_filter_num, blind, stamp_s,
                                cut_frame_num, scan_count, out.ctypes.data, len(out), begin.ctypes.data,
                                offs.ctypes.data, cnts.ctypes.data, cap_f)
    return _unflatten(nf, out, begin, offs, cnts)


_ref_pre = None


def ref_preprocess_lib():
    """The UNMODIFIED reference src/preprocess.cpp built by `make -C oracle ref` (None when it was never built)."""
    global _ref_pre
    if _ref_pre is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libref_preprocess.so")
        if not os.path.exists(path):
            return None
        L = C.CDLL(path)
        L.ref_ingest_pcl2.argtypes = lib().orc_ingest_pcl2.argtypes
        L.ref_ingest_livox.argtypes = lib().orc_ingest_livox.argtypes
        _ref_pre = L
    return _ref_pre


def ref_ingest_pcl2(data, n_points, fields, lidar_type, n_scans, point_filter_num, blind, stamp_s, cut_frame_num, scan_count):
    """Preprocess::process_cut_frame_pcl2 of the reference itself (same signature as ingest_pcl2)."""
    raw = np.ascontiguousarray(np.frombuffer(data, np.uint8))
    f = np.asarray(fields, np.int32)
    out = np.zeros((max(n_points, 1), 4), np.float32)
    cap_f = max(int(cut_frame_num), 1) + 1
    begin, offs, cnts = np.zeros(cap_f), np.zeros(cap_f, np.int32), np.zeros(cap_f, np.int32)
    nf = ref_preprocess_lib().ref_ingest_pcl2(raw.ctypes.data, n_points, f.ctypes.data, lidar_type, n_scans, point_filter_num,
                                              blind, stamp_s, cut_frame_num, scan_count, out.ctypes.data, len(out),
                                              begin.ctypes.data, offs.ctypes.data, cnts.ctypes.data, cap_f)
    return _unflatten(nf, out, begin, offs, cnts)


def ref_ingest_livox(data, n_points, fields, n_scans, point_filter_num, blind, stamp_s, cut_frame_num, scan_count):
    raw = np.ascontiguousarray(np.frombuffer(data, np.uint8))
    f = np.asarray(fields, np.int32)
    out = np.zeros((max(n_points, 1), 4), np.float32)
    cap_f = max(int(cut_frame_num), 1) + 1
    begin, offs, cnts = np.zeros(cap_f), np.zeros(cap_f, np.int32), np.zeros(cap_f, np.int32)
    nf = ref_preprocess_lib().ref_ingest_livox(raw.ctypes.data, n_points, f.ctypes.data, n_scans, point_filter_num, blind,
                                               stamp_s, cut_frame_num, scan_count, out.ctypes.data, len(out),
                                               begin.ctypes.data, offs.ctypes.data, cnts.ctypes.data, cap_f)
    return _unflatten(nf, out, begin, offs, cnts)


_ref_math = None


def ref_math_lib():
    """The UNMODIFIED reference include/so3_math.h + include/common_lib.h (StatesGroup) built by `make -C oracle ref` against
    oracle/ref_shim_math (None when it was never built)."""
    global _ref_math
    if _ref_math is None:
        path = os.path.join(_HERE, "_ref", "libref_math.so")
        if not os.path.exists(path):
            return None
        L = C.CDLL(path)
        D = C.POINTER(C.c_double)
        L.ref_exp1.argtypes = [D, D]
        L.ref_exp_dt.argtypes = [D, C.c_double, D]
        L.ref_exp3.argtypes = [C.c_double, C.c_double, C.c_double, D]
        L.ref_log.argtypes = [D, D]
        L.ref_rot_to_euler.argtypes = [D, D]
        L.ref_skew.argtypes = [D, D]
        L.ref_state_init.argtypes = [D]
        L.ref_state_boxplus.argtypes = [D, D]
        L.ref_state_plus.argtypes = [D, D, D]
        L.ref_state_boxminus.argtypes = [D, D, D]
        L.ref_set_pose6d.argtypes = [C.c_double, D, D, D, D, D, D]
        _ref_math = L
    return _ref_math


class RefMath:
    """numpy front-end of ref_math_lib(): same call shapes as the oracle's exp_so3 / exp3 / log_so3 / rot_to_euler /
    state_init / state_boxplus / state_boxminus."""

    def __init__(self):
        self.L = ref_math_lib()
        if self.L is None:
            raise RuntimeError("oracle/_ref/libref_math.so is not built")

    def exp_so3(self, w, dt=None):
        R, w = np.zeros(9), _f64(w)
        if dt is None:
            self.L.ref_exp1(_dp(w), _dp(R))
        else:
            self.L.ref_exp_dt(_dp(w), C.c_double(dt), _dp(R))
        return R.reshape(3, 3)

    def exp3(self, a, b, c):
        R = np.zeros(9)
        self.L.ref_exp3(a, b, c, _dp(R))
        return R.reshape(3, 3)

    def log_so3(self, R):
        out = np.zeros(3)
        self.L.ref_log(_dp(_f64(R).reshape(-1)), _dp(out))
        return out

    def rot_to_euler(self, R):
        out = np.zeros(3)
        self.L.ref_rot_to_euler(_dp(_f64(R).reshape(-1)), _dp(out))
        return out

    def state_init(self):
        s = np.zeros(STATE_DOUBLES)
        self.L.ref_state_init(_dp(s))
        return s

    def state_boxplus(self, state, d24):
        s = _f64(state).copy()
        self.L.ref_state_boxplus(_dp(s), _dp(_f64(d24)))
        return s

    def state_plus(self, state, d24):
        out = np.zeros(STATE_DOUBLES)
        self.L.ref_state_plus(_dp(_f64(state)), _dp(_f64(d24)), _dp(out))
        return out

    def state_boxminus(self, a, b):
        out = np.zeros(24)
        self.L.ref_state_boxminus(_dp(_f64(a)), _dp(_f64(b)), _dp(out))
        return out

    def set_pose6d(self, t, acc, gyr, vel, pos, R):
        out = np.zeros(22)
        self.L.ref_set_pose6d(t, _dp(_f64(acc)), _dp(_f64(gyr)), _dp(_f64(vel)), _dp(_f64(pos)), _dp(_f64(R).reshape(-1)), _dp(out))
        return out


def num_procs() -> int:
    return lib().orc_num_procs()


# ---------------------------------------------------------------------------------------------------------------------
# The UNMODIFIED reference src/IMU_Processing.hpp (ImuProcess::Process: IMU forward propagation + de-skew, both modes),
# built by `make -C oracle ref` against oracle/ref_shim_imu + oracle/ref_shim_math (oracle/ref_imu_wrap.cpp).
_ref_imu = None


def ref_imu_lib():
    """None when oracle/_ref/libref_imu.so was never built."""
    global _ref_imu
    if _ref_imu is None:
        path = os.path.join(_HERE, "_ref", "libref_imu.so")
        if not os.path.exists(path):
            return None
        L = C.CDLL(path)
        D, F, I = C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_int)
        L.ref_imu_process_lio.restype = C.c_int
        L.ref_imu_process_lio.argtypes = [D, C.c_int, D, C.c_double, D, D, C.c_double, C.c_double, C.c_int, D, F, C.c_int, D, C.c_int, I, D]
        L.ref_imu_process_cv.restype = C.c_int
        L.ref_imu_process_cv.argtypes = [C.c_double, C.c_double, C.c_int, D, C.c_int, D, F, C.c_int]
        _ref_imu = L
    return _ref_imu


def ref_imu_process_lio(imu, last_imu, last_lidar_end_time, acc_s_last, angvel_last, cov_gyr, cov_acc, mean_acc_norm, lidar_beg_time,
                        state_pod, pts4, lidar_type=OUSTER):
    """One LIO-mode ImuProcess::Process call of the reference.  imu: (n, 7) rows (t, gyr, acc); last_imu: (7,).  Returns
    dict(state = propagated lii_state POD, points = de-skewed cloud in the reference's (time-sorted) order, poses = IMUpose table
    (K, 22), acc_s_last, angvel_last, last_lidar_end_time)."""
    L = ref_imu_lib()
    if L is None:
        raise RuntimeError("oracle/_ref/libref_imu.so is not built")
    imu = np.ascontiguousarray(imu, np.float64).reshape(-1, 7)
    st = np.ascontiguousarray(state_pod, np.float64).copy()
    p = _f32(pts4, 4).copy()
    poses = np.zeros((len(imu) + 4, 22))
    n_poses = C.c_int(0)
    carry = np.r_[_f64(acc_s_last), _f64(angvel_last)]
    carry_out = np.zeros(7)
    rc = L.ref_imu_process_lio(_dp(imu), len(imu), _dp(_f64(last_imu)), C.c_double(last_lidar_end_time), _dp(carry),
                               _dp(np.r_[_f64(cov_gyr), _f64(cov_acc)]), C.c_double(mean_acc_norm), C.c_double(lidar_beg_time),
                               int(lidar_type), _dp(st), _fp(p), len(p), _dp(poses), len(poses), C.byref(n_poses), _dp(carry_out))
    if rc != 0:
        raise RuntimeError(f"ref_imu_process_lio failed ({rc})")
    return dict(state=st, points=p, poses=poses[:n_poses.value].copy(), acc_s_last=carry_out[0:3].copy(), angvel_last=carry_out[3:6].copy(),
                last_lidar_end_time=float(carry_out[6]))


def ref_imu_process_cv(lidar_beg_time, time_last_scan, first_frame, cov_gyr_scale, cov_acc_scale, state_pod, pts4, lidar_type=OUSTER):
    """One LO-mode (imu_en = false) ImuProcess::Process call of the reference: constant-velocity propagation + de-skew.  Returns
    (propagated state POD, de-skewed cloud in the reference's time-sorted order)."""
    L = ref_imu_lib()
    if L is None:
        raise RuntimeError("oracle/_ref/libref_imu.so is not built")
    st = np.ascontiguousarray(state_pod, np.float64).copy()
    p = _f32(pts4, 4).copy()
    rc = L.ref_imu_process_cv(C.c_double(lidar_beg_time), C.c_double(time_last_scan), int(bool(first_frame)),
                              _dp(np.r_[_f64(cov_gyr_scale), _f64(cov_acc_scale)]), int(lidar_type), _dp(st), _fp(p), len(p))
    if rc != 0:
        raise RuntimeError(f"ref_imu_process_cv failed ({rc})")
    return st, p


# ------------------------------------------------------------------------------------------------
_ref_iekf = None


def ref_iekf_lib():
    """The reference's OWN TEXT of the per-scan update (src/laserMapping.cpp:936-1134 + map_incremental :516-559, cut out at build
    time by oracle/ref_slice_iekf.py) with the unmodified esti_plane / StatesGroup / ikd-Tree, built by `make -C oracle ref` into
    oracle/_ref/libref_iekf.so (None when it was never built).  ONE map and ONE set of the reference's file-scope variables per process."""
    global _ref_iekf
    if _ref_iekf is None:
        path = os.path.join(_HERE, "_ref", "libref_iekf.so")
        if not os.path.exists(path):
            return None
        L = C.CDLL(path)
        F, D, I, U = C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_ubyte)
        L.ref_iekf_map_build.argtypes = [F, C.c_int, C.c_double]
        L.ref_iekf_update.argtypes = [F, C.c_int, D, C.c_int, C.c_int, I, I, I, U, F, F, I]
        L.ref_iekf_map_incremental.argtypes = [I]
        L.ref_iekf_tree_flatten.argtypes = [F, C.c_int, C.c_int]
        L.ref_esti_plane.argtypes = [F, C.c_double, D]
        _ref_iekf = L
    return _ref_iekf


class RefIekf:
    """numpy front-end of ref_iekf_lib(): map_build(xyz, filter_size_map); update(body, state) -> dict shaped like Tree.iekf_update's
    (state, iters, rematch, effect_num, nearest, nearest_n, selected, normvec); map_incremental() -> (add_point_size, tree size);
    flatten()."""

    def __init__(self):
        self.L = ref_iekf_lib()
        assert self.L is not None, "oracle/_ref/libref_iekf.so not built (make -C oracle ref, needs /root/reference)"

    def map_build(self, xyz, filter_size_map):
        xyz = _f32(xyz, 3)
        return self.L.ref_iekf_map_build(_fp(xyz), len(xyz), float(filter_size_map))

    def update(self, body, state, max_iterations=4, imu_en=False):
        body = _f32(np.asarray(body)[:, :3], 3)
        n = len(body)
        st = _f64(state).copy()
        it, rm, eff = C.c_int(0), C.c_int(0), C.c_int(0)
        sel = np.zeros(n, np.uint8)
        normvec = np.zeros((n, 4), np.float32)
        near = np.zeros((n, 5, 3), np.float32)
        nn = np.zeros(n, np.int32)
        rc = self.L.ref_iekf_update(_fp(body), n, _dp(st), max_iterations, int(imu_en), C.byref(it), C.byref(rm), C.byref(eff),
                                    sel.ctypes.data_as(C.POINTER(C.c_ubyte)), _fp(normvec), _fp(near), _ip(nn))
        assert rc == 0, "more than 100 000 points: the reference's arrays end there (quirk A1)"
        return dict(state=st, iters=it.value, rematch=rm.value, effect_num=eff.value, selected=sel, normvec=normvec, nearest=near,
                    nearest_n=nn)

    def esti_plane(self, pts5x3, threshold=0.1):
        pts = _f32(np.asarray(pts5x3).reshape(5, 3), 3)
        out = np.zeros(4)
        ok = self.L.ref_esti_plane(_fp(pts), float(threshold), _dp(out))
        return bool(ok), out

    def map_incremental(self):
        ts = C.c_int(0)
        added = self.L.ref_iekf_map_incremental(C.byref(ts))
        return added, ts.value

    def flatten(self, settle_ms=200, cap=4_000_000):
        out = np.zeros((cap, 3), np.float32)
        n = self.L.ref_iekf_tree_flatten(_fp(out), cap, settle_ms)
        return out[:n].copy()
