"""CPU-side checks of the drop-in boundary: the shared library loads, exports every symbol that
include/liinit_hip.h declares, and refuses to run without a GPU (no silent CPU fallback)."""
import ctypes
import os
import re

import pytest

import lidar_imu_init_amd as lii
from lidar_imu_init_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "liinit_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(lii_[a-z0-9_]+)\s*\(", hdr)))


def test_header_symbols_all_exported():
    assert os.path.exists(lii.library_path()), "build libliinit_hip.so first (__graft_entry__.build())"
    L = ctypes.CDLL(lii.library_path())
    names = _declared_symbols()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, f"declared in include/liinit_hip.h but not exported: {missing}"


def test_python_mirror_covers_header():
    assert sorted(api.EXPORTED_SYMBOLS) == _declared_symbols()
    L = lii.load_library()
    import re
    header = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "liinit_hip.h")).read()
    assert L.lii_abi_version() == int(re.search(r"#define\s+LII_ABI_VERSION\s+(\d+)", header).group(1)) == 9


def test_struct_layouts_match_header():
    assert ctypes.sizeof(api.lii_config) == 48
    assert ctypes.sizeof(api.lii_iekf_report) == 16 + 91 * 8
    assert api.STATE_DOUBLES == 612


def test_fails_loudly_without_gpu():
    L = lii.load_library()
    n = ctypes.c_int(0)
    rc = L.lii_device_count(ctypes.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip("a GPU is visible here")
    with pytest.raises(lii.LIIError):
        lii.Registrar(1000, 1000)
    cfg = api.lii_config()
    cfg.struct_size = ctypes.sizeof(api.lii_config)
    cfg.max_scan_points = 10
    cfg.max_map_points = 10
    h = ctypes.c_void_p()
    assert L.lii_create(ctypes.byref(cfg), ctypes.byref(h)) == -2  # LII_ERR_NO_DEVICE
    assert not h.value


def test_cxx_host_loop_links_against_the_boundary():
    """harness/libliinit_stream.so (the C++ per-scan loop bench.py times) is built from include/liinit_hip.h alone, resolves
    the product library by name and rejects a null handle without touching a device."""
    path = os.path.join(ROOT, "harness", "libliinit_stream.so")
    assert os.path.exists(path), "build it with __graft_entry__.build() (make -C harness)"
    lii.load_library()  # the same libliinit_hip.so the driver library resolves through its rpath
    drv = ctypes.CDLL(path)
    drv.lii_stream_run.restype = ctypes.c_int
    drv.lii_stream_run.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                   ctypes.c_float, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                   ctypes.c_void_p, ctypes.c_void_p]
    totals = (ctypes.c_int64 * 2)()
    assert drv.lii_stream_run(None, None, 0, 0, 0, 0.0, 5, 1, 0, 0, totals, None) == -1  # LII_ERR_INVALID
    assert ctypes.sizeof(api.lii_scan_job) == 88  # struct_size the C++ loop fills in (ABI 9; 72: ABI 8; 56: ABI 6 - 7, scan_sorted behind n_scan_dev)
    # ... and map_update in the formerly reserved word behind it (a job that leaves it 0 behaves as before)
    assert api.lii_scan_job.scan_sorted.offset == 44 and api.lii_scan_job.map_update.offset == 48
    import re
    hdr = open(os.path.join(ROOT, "include", "liinit_hip.h")).read()
    body = hdr[hdr.index("typedef struct lii_scan_job {"):hdr.index("} lii_scan_job;")]
    fields = re.findall(r"^\s*(?:const\s+)?[A-Za-z_0-9]+\*?\s+\*?([a-z_0-9]+);", body, re.M)
    assert fields[-6:] == ["scan_sorted", "map_update", "next_scan_dev", "next_n_scan", "reserved1", "while_waiting_arg"], fields
    # ABI 9: the hook of the host's time inside the call sits behind the job of ABI 8
    assert api.lii_scan_job.while_waiting.offset == 72 and api.lii_scan_job.while_waiting_arg.offset == 80
    # ABI 8: the announcement of the next scan sits BEHIND the job of ABI 6 - 7 (a caller that fills in 56 bytes and says so is served as before)
    assert api.lii_scan_job.next_scan_dev.offset == 56 and api.lii_scan_job.next_n_scan.offset == 64
