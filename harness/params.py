"""Workload parameters of bench.py / the harnesses: read from reference-format yaml + launch files (harness/config,
harness/launch) through the library's loader (lii_params_load_launch) - the same way a host would read the reference's own
config/*.yaml and launch/*.launch (src/laserMapping.cpp:767-799)."""
import os

HERE = os.path.dirname(os.path.abspath(__file__))
ALIASES = {"ouster": "os1_128", "velodyne": "vlp16", "hesai": "hesai128", "avia": "avia"}


def load(name, **overrides):
    from lidar_imu_init_amd.params import Params
    name = ALIASES.get(name, name)
    return Params(launch=os.path.join(HERE, "launch", name + ".launch"), config_dir=os.path.join(HERE, "config"), **overrides)
