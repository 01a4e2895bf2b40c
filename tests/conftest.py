import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def small_world():
    """A 24 x 18 x 6 m hall, its 0.15 m-lattice map (~70 k points) and a 2 k-point scan."""
    from harness import synth
    hall = synth.Hall(size=(24.0, 18.0, 6.0), n_boxes=8, seed=7)
    map_pts = hall.surface_points(0.15, noise=0.01, seed=7)
    return hall, map_pts


def make_state(oracle, R=None, p=None, R_LI=None, T_LI=None):
    st = oracle.state_init()
    v = oracle.StateView(st)
    if R is not None:
        v.rot_end[:] = R
    if p is not None:
        v.pos_end[:] = p
    if R_LI is not None:
        v.offset_R_L_I[:] = R_LI
    if T_LI is not None:
        v.offset_T_L_I[:] = T_LI
    return st
