"""The parameter surface (config/*.yaml + launch/*.launch, src/laserMapping.cpp:767-799) through the C-ABI loader
(lii_params_*): defaults = the nh.param<> defaults, every file of the harness and - where the reference tree is present -
every yaml and launch file the reference ships round-trips into lii_config / lii_ingest_opts / lii_iekf_opts; an independent
parse of the same files (PyYAML) is the checker.  Host code only: runs without a GPU."""
import ctypes as C
import glob
import os
import re

import pytest

from lidar_imu_init_amd import api
from lidar_imu_init_amd.params import Params, lii_params

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

# nh.param name -> (field, default) exactly as src/laserMapping.cpp:770-799 lists them
NAMES = {
    "max_iteration": ("max_iteration", 4), "point_filter_num": ("point_filter_num", 2),
    "common/lid_topic": ("lid_topic", "/livox/lidar"), "common/imu_topic": ("imu_topic", "/livox/imu"),
    "mapping/filter_size_surf": ("filter_size_surf", 0.5), "mapping/filter_size_map": ("filter_size_map", 0.5),
    "cube_side_length": ("cube_side_length", 200.0), "mapping/det_range": ("det_range", 300.0),
    "mapping/gyr_cov": ("gyr_cov", 0.1), "mapping/acc_cov": ("acc_cov", 0.1), "mapping/grav_cov": ("grav_cov", 0.001),
    "mapping/b_gyr_cov": ("b_gyr_cov", 0.0001), "mapping/b_acc_cov": ("b_acc_cov", 0.0001),
    "preprocess/blind": ("blind", 1.0), "preprocess/lidar_type": ("lidar_type", 1), "preprocess/scan_line": ("scan_line", 16),
    "preprocess/feature_extract_en": ("feature_extract_en", 0), "initialization/cut_frame": ("cut_frame", 1),
    "initialization/cut_frame_num": ("cut_frame_num", 1), "initialization/orig_odom_freq": ("orig_odom_freq", 10),
    "initialization/online_refine_time": ("online_refine_time", 20.0), "initialization/mean_acc_norm": ("mean_acc_norm", 9.81),
    "initialization/data_accum_length": ("data_accum_length", 300.0),
    "publish/path_en": ("path_en", 1), "publish/scan_publish_en": ("scan_publish_en", 1),
    "publish/dense_publish_en": ("dense_publish_en", 1), "publish/scan_bodyframe_pub_en": ("scan_bodyframe_pub_en", 1),
    "runtime_pos_log_enable": ("runtime_pos_log_enable", 0), "pcd_save/pcd_save_en": ("pcd_save_en", 0),
    "pcd_save/interval": ("pcd_save_interval", -1),
}


def _flatten(d, prefix=""):
    out = {}
    for k, v in (d or {}).items():
        if isinstance(v, dict):
            out.update(_flatten(v, prefix + k + "/"))
        else:
            out[prefix + k] = v
    return out


def _expected(yaml_path, launch_path=None):
    import yaml
    exp = {f: d for f, d in NAMES.values()}
    exp["Rot_LI_cov"], exp["Trans_LI_cov"] = None, None
    flat = _flatten(yaml.safe_load(open(yaml_path)))
    if launch_path:
        xml = re.sub(r"<!--.*?-->", "", open(launch_path).read(), flags=re.S)
        for m in re.finditer(r"<param\s+([^>]*)/?>", xml):
            attrs = dict(re.findall(r'(\w+)\s*=\s*"([^"]*)"', m.group(1)))
            if "name" in attrs and "value" in attrs:
                flat[attrs["name"]] = attrs["value"]
    for name, v in flat.items():
        if name in NAMES:
            f, d = NAMES[name]
            exp[f] = type(d)(float(v)) if isinstance(d, (int, float)) and not isinstance(v, bool) else (int(v) if isinstance(v, bool) else v)
        elif name in ("initialization/Rot_LI_cov", "initialization/Trans_LI_cov"):
            exp[name.split("/")[1]] = [float(x) for x in v]
    return exp


def _compare(p, exp):
    for f, v in exp.items():
        got = getattr(p, f)
        if v is None:
            continue
        if isinstance(v, list):
            assert got[:len(v)] == v, f
        else:
            assert got == v, (f, got, v)


def test_defaults_are_the_nh_param_defaults():
    p = Params()
    assert p.struct_size == C.sizeof(lii_params)
    for name, (f, d) in NAMES.items():
        assert getattr(p, f) == d, name
    assert p.n_Rot_LI_cov == 0 and p.n_Trans_LI_cov == 0 and p.map_file_path == ""


@pytest.mark.parametrize("name", ["os1_128", "vlp16", "hesai128", "avia"])
def test_harness_configs_round_trip(name):
    y = os.path.join(ROOT, "harness", "config", name + ".yaml")
    l = os.path.join(ROOT, "harness", "launch", name + ".launch")
    p = Params(launch=l, config_dir=os.path.dirname(y))
    _compare(p, _expected(y, l))
    cfg, ing, opts, leaf = p.apply(device=0, max_scan_points=1234, max_map_points=5678)
    assert cfg.struct_size == C.sizeof(api.lii_config) and cfg.max_scan_points == 1234 and cfg.max_map_points == 5678
    assert cfg.map_downsample_size == pytest.approx(p.filter_size_map) and cfg.max_match_dist2 == 5.0
    assert cfg.plane_threshold == 0.1 and cfg.laser_point_cov_inv == 1000.0
    assert ing.struct_size == C.sizeof(api.lii_ingest_opts)
    assert (ing.lidar_type, ing.n_scans, ing.point_filter_num, ing.cut_frame_num) == (p.lidar_type, p.scan_line, p.point_filter_num, p.cut_frame_num)
    assert ing.blind == p.blind and opts.max_iterations == p.max_iteration and opts.imu_en == 0
    assert leaf == pytest.approx(p.filter_size_surf)


def test_overrides_and_errors(tmp_path):
    p = Params(yaml=os.path.join(ROOT, "harness", "config", "os1_128.yaml"))
    assert p.max_iteration == 4 and p.scan_line == 128  # the launch file was not read: nh.param default
    p.set("max_iteration", 7).set("/initialization/cut_frame_num", 3).set("initialization/cut_frame", False)
    p.set("initialization/Trans_LI_cov", [1e-3, 2e-3, 3e-3]).set("a/name/nobody/reads", 1)
    assert p.max_iteration == 7 and p.cut_frame_num == 3 and p.cut_frame == 0 and p.Trans_LI_cov == [1e-3, 2e-3, 3e-3]
    assert p.apply()[1].cut_frame_num == 0  # cut_frame false: Preprocess::process instead of process_cut_frame_* (laserMapping.cpp:326-342, :363-379)
    with pytest.raises(api.LIIError):
        p.set("mapping/filter_size_surf", "wide")
    with pytest.raises(api.LIIError):
        Params(yaml=str(tmp_path / "missing.yaml"))
    bad = tmp_path / "bad.yaml"
    bad.write_text("mapping:\n  filter_size_map: 0\n")
    with pytest.raises(api.LIIError):
        Params(yaml=str(bad)).apply()
    odd = tmp_path / "odd.yaml"  # comments, quotes, tabs after values, CRLF, a '#' inside a quoted string, deeper nesting
    odd.write_text("common:\r\n    lid_topic:  '/a#b'   # trailing\r\n    imu_topic: \"/imu # not a comment\"\r\n"
                   "mapping:\n\tfilter_size_surf: 0.25\t# tab indent\n  extra:\n    deeper: {}\npreprocess: \n  blind: 4e-1\n")
    q = Params(yaml=str(odd))
    assert q.lid_topic == "/a#b" and q.imu_topic == "/imu # not a comment" and q.filter_size_surf == 0.25 and q.blind == 0.4
    launch = tmp_path / "x.launch"
    launch.write_text('<launch>\n<!-- <param name="max_iteration" value="99"/> -->\n<rosparam command="load" file="$(find pkg)/config/odd.yaml"/>\n'
                      '<param name="max_iteration" type="int"\n   value="6" />\n<node pkg="p" type="t" name="n"/>\n</launch>\n')
    r = Params(launch=str(launch), config_dir=str(tmp_path))
    assert r.max_iteration == 6 and r.filter_size_surf == 0.25
    assert Params(launch=str(launch)).filter_size_surf == 0.25  # falls back to the launch file's own directory
    (tmp_path / "sub").mkdir()
    (tmp_path / "sub" / "y.launch").write_text(launch.read_text().replace("odd.yaml", "nowhere.yaml"))
    with pytest.raises(api.LIIError):
        Params(launch=str(tmp_path / "sub" / "y.launch"))  # the yaml cannot be resolved


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "config")), reason="reference tree not present")
def test_every_shipped_reference_file_loads():
    yamls = sorted(glob.glob(os.path.join(REF, "config", "*.yaml")))
    launches = sorted(glob.glob(os.path.join(REF, "launch", "*.launch")))
    assert len(yamls) >= 7 and len(launches) >= 7
    for y in yamls:
        _compare(Params(yaml=y), _expected(y))
    for l in launches:
        xml = open(l).read()
        y = os.path.join(REF, "config", re.search(r"config/(\w+\.yaml)", xml).group(1))
        p = Params(launch=l)  # resolves <launch dir>/../config/X.yaml
        _compare(p, _expected(y, l))
        p.apply()
