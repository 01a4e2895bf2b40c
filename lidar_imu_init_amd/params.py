"""ctypes mirror of the parameter surface of include/liinit_hip.h (lii_params_*): the reference's config/*.yaml +
launch/*.launch names (src/laserMapping.cpp:767-799).  Parsing happens in the C++ library; nothing is parsed in Python."""
from __future__ import annotations

import ctypes as C

from . import api


class lii_params(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("max_iteration", C.c_int32), ("point_filter_num", C.c_int32),
                ("filter_size_surf", C.c_double), ("filter_size_map", C.c_double), ("cube_side_length", C.c_double),
                ("det_range", C.c_double), ("gyr_cov", C.c_double), ("acc_cov", C.c_double), ("grav_cov", C.c_double),
                ("b_gyr_cov", C.c_double), ("b_acc_cov", C.c_double), ("blind", C.c_double), ("lidar_type", C.c_int32),
                ("scan_line", C.c_int32), ("feature_extract_en", C.c_int32), ("cut_frame", C.c_int32),
                ("cut_frame_num", C.c_int32), ("orig_odom_freq", C.c_int32), ("online_refine_time", C.c_double),
                ("mean_acc_norm", C.c_double), ("data_accum_length", C.c_double), ("Rot_LI_cov", C.c_double * 3),
                ("Trans_LI_cov", C.c_double * 3), ("n_Rot_LI_cov", C.c_int32), ("n_Trans_LI_cov", C.c_int32),
                ("path_en", C.c_int32), ("scan_publish_en", C.c_int32), ("dense_publish_en", C.c_int32),
                ("scan_bodyframe_pub_en", C.c_int32), ("runtime_pos_log_enable", C.c_int32), ("pcd_save_en", C.c_int32),
                ("pcd_save_interval", C.c_int32), ("reserved0", C.c_int32), ("lid_topic", C.c_char * 128),
                ("imu_topic", C.c_char * 128), ("map_file_path", C.c_char * 256)]


def _lib():
    return api.load_library()


def _check(rc):
    if rc != 0:
        raise api.LIIError(rc, (_lib().lii_params_last_error() or b"").decode())


class Params:
    """One parameter set: defaults -> yaml -> launch overrides -> explicit overrides; attribute access reads the POD."""

    def __init__(self, yaml: str | None = None, launch: str | None = None, config_dir: str | None = None, **overrides):
        self.pod = lii_params()
        _check(_lib().lii_params_defaults(C.byref(self.pod)))
        if yaml:
            self.load_yaml(yaml)
        if launch:
            self.load_launch(launch, config_dir)
        for k, v in overrides.items():
            self.set(k.replace("__", "/"), v)

    def load_yaml(self, path):
        _check(_lib().lii_params_load_yaml(str(path).encode(), C.byref(self.pod)))
        return self

    def load_launch(self, path, config_dir=None):
        _check(_lib().lii_params_load_launch(str(path).encode(), str(config_dir).encode() if config_dir else None, C.byref(self.pod)))
        return self

    def set(self, name, value):
        if isinstance(value, bool):
            value = "true" if value else "false"
        elif isinstance(value, (list, tuple)):
            value = "[" + ", ".join(repr(float(x)) for x in value) + "]"
        _check(_lib().lii_params_set(C.byref(self.pod), name.encode(), str(value).encode()))
        return self

    def __getattr__(self, name):
        pod = object.__getattribute__(self, "pod")
        v = getattr(pod, name)
        if isinstance(v, bytes):
            return v.decode()
        if hasattr(v, "__len__"):
            return list(v)
        return v

    def apply(self, device=0, max_scan_points=200_000, max_map_points=2_000_000):
        """-> (lii_config, lii_ingest_opts, lii_iekf_opts, leaf) as the C-ABI fills them."""
        cfg, ing, opts, leaf = api.lii_config(), api.lii_ingest_opts(), api.lii_iekf_opts(), C.c_float(0)
        _check(_lib().lii_params_apply(C.byref(self.pod), device, max_scan_points, max_map_points, C.byref(cfg), C.byref(ing),
                                       C.byref(opts), C.byref(leaf)))
        return cfg, ing, opts, leaf.value
