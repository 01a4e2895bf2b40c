"""oracle/orc_math.hpp held BIT FOR BIT to the reference's own include/so3_math.h and include/common_lib.h:68-169
(StatesGroup), compiled UNMODIFIED from /root/reference against a minimal fixed-size-matrix shim (oracle/ref_shim_math,
oracle/ref_math_wrap.cpp -> oracle/_ref/libref_math.so; built here by `make -C oracle ref`, carried to the GPU box prebuilt).

Covers every branch of the reference functions: Exp(ang) / Exp(ang_vel, dt) around their 1e-7 identity threshold, Exp(v1, v2, v3)
around 1e-5, Log around trace > 3 - 1e-6 and |theta| < 1e-3, RotMtoEuler on both sides of its singularity, StatesGroup() (initial
covariance), operator+= / operator+ / operator-."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def ref(oracle):
    if oracle.ref_math_lib() is None:
        oracle.build()
    if oracle.ref_math_lib() is None:
        pytest.skip("oracle/_ref/libref_math.so not built (no /root/reference here and no prebuilt copy)")
    return oracle.RefMath()


def _vectors(rng):
    out = [np.zeros(3), np.array([1e-8, 0, 0]), np.array([0, 9.9e-8, 0]), np.array([1.0000001e-7, 0, 0]), np.array([6e-8, 6e-8, 6e-8]),
           np.array([9e-6, 0, 0]), np.array([0, 0, 1.1e-5]), np.array([6e-6, 6e-6, 6e-6]), np.array([np.pi, 0, 0]), np.array([2.0, -2.0, 1.5])]
    for scale in (1e-9, 1e-7, 1e-5, 1e-3, 1e-1, 1.0, 3.0):
        out += list(rng.normal(0, scale, (40, 3)))
    return out


def test_exp_log_euler_bit_exact(oracle, ref):
    rng = np.random.default_rng(7)
    for w in _vectors(rng):
        assert np.array_equal(oracle.exp_so3(w), ref.exp_so3(w)), w
        for dt in (1.0, 0.01, -0.037, 1e-4):
            assert np.array_equal(oracle.exp_so3(w, dt), ref.exp_so3(w, dt)), (w, dt)
        assert np.array_equal(oracle.exp3(*w), ref.exp3(*w)), w
        R = ref.exp_so3(w)
        assert np.array_equal(oracle.log_so3(R), ref.log_so3(R)), w          # incl. trace > 3 - 1e-6 and |theta| < 1e-3
        assert np.array_equal(oracle.rot_to_euler(R), ref.rot_to_euler(R)), w
    # Log of matrices that are only approximately rotations (what accumulates in the filter) and the euler singularity
    for _ in range(200):
        M = ref.exp_so3(rng.normal(0, 1.0, 3)) + rng.normal(0, 1e-9, (3, 3))
        assert np.array_equal(oracle.log_so3(M), ref.log_so3(M))
        assert np.array_equal(oracle.rot_to_euler(M), ref.rot_to_euler(M))
    for pitch in (np.pi / 2, -np.pi / 2, np.pi / 2 - 1e-7):
        c, s = np.cos(pitch), np.sin(pitch)
        Ry = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]]) @ ref.exp_so3(np.array([0.3, 0, 0]))
        assert np.array_equal(oracle.rot_to_euler(Ry), ref.rot_to_euler(Ry))


def test_statesgroup_bit_exact(oracle, ref):
    assert np.array_equal(oracle.state_init(), ref.state_init())             # INIT_COV 1, last 9 diagonal 1e-5 (common_lib.h:79-80)
    rng = np.random.default_rng(11)
    for k in range(200):
        s = ref.state_init()
        s[0:9] = ref.exp_so3(rng.normal(0, 1.0, 3)).reshape(-1)
        s[12:21] = ref.exp_so3(rng.normal(0, 0.5, 3)).reshape(-1)
        s[9:12], s[21:24] = rng.normal(0, 10, 3), rng.normal(0, 0.2, 3)
        s[24:36] = rng.normal(0, 1, 12)
        A = rng.normal(0, 1, (24, 24))
        s[36:] = (A @ A.T).reshape(-1)
        scale = [1e-9, 1e-6, 1e-3, 0.3][k % 4]
        d = rng.normal(0, scale, 24)
        if k % 7 == 0:
            d[0:3] = [3e-6, 0, 0]   # rotation increment below the 1e-5 identity threshold of Exp(v1, v2, v3)
            d[6:9] = 0
        a = ref.state_boxplus(s, d)
        assert np.array_equal(oracle.state_boxplus(s, d), a)                  # operator+= (the update's `state += solution`)
        assert np.array_equal(ref.state_plus(s, d), a)                        # operator+ agrees with += (cov carried over)
        assert np.array_equal(oracle.state_boxminus(a, s), ref.state_boxminus(a, s))   # operator- (state_propagat - state)
        assert np.array_equal(oracle.state_boxminus(s, a), ref.state_boxminus(s, a))


def test_set_pose6d_layout(ref):
    rng = np.random.default_rng(3)
    acc, gyr, vel, pos = rng.normal(0, 1, (4, 3))
    R = ref.exp_so3(rng.normal(0, 1, 3))
    k = ref.set_pose6d(0.0125, acc, gyr, vel, pos, R)
    # offset_time, acc, gyr, vel, pos, rot row-major: the 22-double record of lii_pose6d (include/liinit_hip.h)
    assert np.array_equal(k, np.r_[0.0125, acc, gyr, vel, pos, R.reshape(-1)])
