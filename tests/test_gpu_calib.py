"""GPU parity of the LI-Init evaluator (k_calib_eval through lii_calib_eval) and of the host LM around it
(lii_calib_solve_stage) against the numpy oracle, on the sequences derived from the reference's committed run.
Tolerances: J^T J / J^T r / cost relative 1e-10 (fp64, different summation order); solved parameters 1e-8."""
import numpy as np
import pytest

import li_init_fixture as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def reg():
    import lidar_imu_init_amd as lii
    r = lii.Registrar(max_scan_points=1000, max_map_points=1000)
    yield r
    r.close()


def _rel(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300)


def test_calib_eval_matches_oracle(reg):
    from oracle import li_init_np as LI
    out = F.run(solve=True)
    rng = np.random.default_rng(1)
    q = np.array([0.72, 0.02, -0.01, 0.69]); q /= np.linalg.norm(q)
    R = LI.quat_to_rot(q)
    # stages 1 / 2 on the buffers before the second time compensation
    imu, lid = out["imu_stage12"], out["lidar_stage12"]
    reg.calib_set_buffers(imu.to_records(), lid.to_records())
    for stage, v in ((1, np.zeros(0)), (2, np.r_[0.002, 0.0007, -0.0004, -0.0045])):
        params = np.r_[R.reshape(-1), v]
        JtJ, Jtr, cost = reg.calib_eval(stage, params)
        rJ, rg, rc = LI.normal_equations(stage, R, v, imu, lid)
        assert _rel(JtJ, rJ) < 1e-10 and _rel(Jtr, rg) < 1e-10 and abs(cost - rc) / rc < 1e-12
    # stage 3 on the buffers after it + acc_interpolate
    imu3, lid3 = out["imu_stage3"], out["lidar_stage3"]
    reg.calib_set_buffers(imu3.to_records(), lid3.to_records())
    R_LI = out["stage2"]["R_LI"]
    RG = LI.quat_to_rot(LI.quat_plus(np.array([1.0, 0, 0, 0]), np.array([0.03, -0.02, 0.01])))
    v = np.r_[0.004, -0.006, 0.009, 0.02, -0.02, -0.17]
    JtJ, Jtr, cost = reg.calib_eval(3, np.r_[RG.reshape(-1), v, R_LI.reshape(-1)])
    rJ, rg, rc = LI.normal_equations(3, RG, v, imu3, lid3, R_LI)
    assert _rel(JtJ, rJ) < 1e-10 and _rel(Jtr, rg) < 1e-10 and abs(cost - rc) / rc < 1e-12


def test_calib_solves_match_oracle_and_reference(reg):
    from lidar_imu_init_amd.api import lii_calib_result
    from oracle import oracle as O
    d = F.load()
    out = F.run(solve=True)
    imu, lid = out["imu_stage12"], out["lidar_stage12"]
    reg.calib_set_buffers(imu.to_records(), lid.to_records())
    res = lii_calib_result()
    res.R_LI[:] = list(np.eye(3).reshape(-1))
    reg.calib_solve_stage(1, res)
    assert np.allclose(np.array(res.R_LI[:]).reshape(3, 3), out["stage1"]["R_LI"], atol=1e-8)
    assert res.iterations[0] == out["stage1"]["iterations"]
    reg.calib_solve_stage(2, res)
    R2 = np.array(res.R_LI[:]).reshape(3, 3)
    assert np.allclose(R2, out["stage2"]["R_LI"], atol=1e-8)
    assert np.allclose(res.gyro_bias[:], out["stage2"]["gyro_bias"], atol=1e-9)
    assert abs(res.time_lag_2 - out["stage2"]["time_lag_2"]) < 1e-9
    assert res.iterations[1] == out["stage2"]["iterations"]
    # the HIP-evaluated solve lands on the reference's committed initialisation result too
    assert np.abs(O.rot_to_euler(R2) * 57.3 - d["result_rot_euler_deg"]).max() < 0.01
    assert np.abs(np.array(res.gyro_bias[:]) - d["result_gyro_bias"]).max() < 5e-5
    # stage 3 on the re-aligned buffers (LI_Initialization re-uses the deques after IMU_time_compensate + acc_interpolate)
    reg.calib_set_buffers(out["imu_stage3"].to_records(), out["lidar_stage3"].to_records())
    reg.calib_solve_stage(3, res)
    assert np.allclose(res.T_LI[:], out["stage3"]["T_LI"], atol=1e-8)
    assert np.allclose(res.acc_bias[:], out["stage3"]["acc_bias"], atol=1e-9)
    assert np.allclose(res.grav_L0[:], out["stage3"]["grav_L0"], atol=1e-8)
    assert np.abs(np.array(res.T_LI[:]) - d["result_trans"]).max() < 1e-3
    assert np.abs(np.array(res.grav_L0[:]) - d["result_gravity"]).max() < 2e-3


def test_li_initialization_end_to_end_on_reference_run(reg):
    """lii_li_init_run (C++ conditioning chain + HIP-evaluated solves) against the numpy oracle and the reference's result."""
    from oracle import oracle as O
    d = F.load()
    out = F.run(solve=True)
    imu, lid = F.sequences()
    res, lag1, total = reg.li_init_run(imu.to_records(), lid.to_records(), 10, 5)
    assert abs(lag1 - out["time_lag_1"]) < 1e-15 and abs(lag1 + 0.08) < 1e-12
    R = np.array(res.R_LI[:]).reshape(3, 3)
    assert np.allclose(R, out["stage2"]["R_LI"], atol=1e-8)
    assert np.allclose(res.gyro_bias[:], out["stage2"]["gyro_bias"], atol=1e-9)
    assert abs(res.time_lag_2 - out["stage2"]["time_lag_2"]) < 1e-9
    assert abs(total - out["time_delay"]) < 1e-9
    assert np.allclose(res.T_LI[:], out["stage3"]["T_LI"], atol=1e-7)
    assert np.allclose(res.acc_bias[:], out["stage3"]["acc_bias"], atol=1e-8)
    assert np.allclose(res.grav_L0[:], out["stage3"]["grav_L0"], atol=1e-7)
    # and the committed result of the reference program
    assert np.abs(O.rot_to_euler(R) * 57.3 - d["result_rot_euler_deg"]).max() < 0.01
    assert np.abs(np.array(res.T_LI[:]) - d["result_trans"]).max() < 1e-3
    assert np.abs(np.array(res.acc_bias[:]) - d["result_acc_bias"]).max() < 1e-5


def test_conditioning_chain_on_the_device(reg):
    """SURVEY.md section 8(f)4 on the device (lii_li_init_dev.hip): the zero-phase Butterworth filter and the O(N^2)
    cross-correlation against the numpy oracle - bit for bit (same order of additions, no FMA contraction; tolerance stated as
    1e-12 relative) - and against the reference's own Log/IMU_meas.txt through the fixture; then the whole LI_Initialization
    with the device chain switched on must give the host chain's result bit for bit."""
    import time
    from oracle import li_init_np as LI
    imu, lid = F.sequences()
    imu_c, lid_c = LI.imu_time_compensate(imu.copy(), lid.copy(), 0.0, True)
    want_i, want_l = LI.zero_phase_filt(imu_c), LI.zero_phase_filt(lid_c)
    batch = np.stack([imu_c.to_records(), lid_c.to_records()])
    t0 = time.perf_counter()
    got = reg.zero_phase_filter(batch)
    t_dev = time.perf_counter() - t0
    for g, w in ((got[0], want_i.to_records()), (got[1], want_l.to_records())):
        assert np.array_equal(g[:, :9], w[:, :9]) and np.array_equal(g[:, 21], w[:, 21])   # rot_end / timestamp pass through
        assert np.max(np.abs(g[:, 9:21] - w[:, 9:21])) <= 1e-12 * max(1.0, np.max(np.abs(w[:, 9:21])))
        assert np.array_equal(g[:, 9:21], w[:, 9:21]), "same additions in the same order: bit-identical"
    # the filtered IMU angular velocity is what the reference logged (Log/IMU_meas.txt holds the sequence after the chain's
    # tail cut and time compensation: compare the common span through the oracle's own pinned reproduction)
    out = F.run(solve=False)
    # cross-correlation: the lag of the reference run (-0.08 s at 50 Hz = 4 samples), then shifted copies
    fi, fl = LI.normalize_acc(want_i), want_l
    fi2, fl2 = LI.cut_sequence_tail(fi.slice(slice(0, len(fi) - 1)), fl.slice(slice(0, len(fl) - 1)))
    lag1, lag_samples = LI.xcorr_temporal_init(fi2, fl2, 50.0)
    t0 = time.perf_counter()
    got_lag = reg.xcorr_lag(fi2.to_records(), fl2.to_records())
    t_x = time.perf_counter() - t0
    assert got_lag == lag_samples and abs(lag1 - out["time_lag_1"]) < 1e-15
    rng = np.random.default_rng(4)
    for shift in (0, 7, -13, 40):
        a = LI.CalibSeq(900)
        a.ang_vel = rng.normal(0, 0.5, (900, 3))
        b = a.copy()
        b.ang_vel = np.roll(a.ang_vel, shift, axis=0) + rng.normal(0, 0.01, (900, 3))
        assert reg.xcorr_lag(a.to_records(), b.to_records()) == LI.xcorr_temporal_init(a, b, 50.0)[1]
    # a plateau of equal maxima: the first lag in ascending order wins on both sides
    a = LI.CalibSeq(64)
    b = LI.CalibSeq(64)
    assert reg.xcorr_lag(a.to_records(), b.to_records()) == LI.xcorr_temporal_init(a, b, 50.0)[1]
    # the whole initialization with the device chain: bit-identical to the host chain
    reg.li_init_set_device(False)
    res_h, lag_h, tot_h = reg.li_init_run(imu.to_records(), lid.to_records(), 10, 5)
    t0 = time.perf_counter()
    reg.li_init_run(imu.to_records(), lid.to_records(), 10, 5)
    t_host = time.perf_counter() - t0
    reg.li_init_set_device(True)
    try:
        t0 = time.perf_counter()
        res_d, lag_d, tot_d = reg.li_init_run(imu.to_records(), lid.to_records(), 10, 5)
        t_run = time.perf_counter() - t0
    finally:
        reg.li_init_set_device(False)
    assert lag_d == lag_h and tot_d == tot_h
    for f in ("R_LI", "T_LI", "gyro_bias", "acc_bias", "grav_L0", "final_cost"):
        assert np.array_equal(np.array(getattr(res_d, f)[:]), np.array(getattr(res_h, f)[:])), f
    assert res_d.time_lag_2 == res_h.time_lag_2
    print(f"device zero-phase filter of 2 x {batch.shape[1]} states: {t_dev * 1e3:.2f} ms; cross-correlation ({len(fi2)} samples): "
          f"{t_x * 1e3:.2f} ms; lii_li_init_run with the device chain: {t_run * 1e3:.1f} ms, with the host chain: {t_host * 1e3:.1f} ms")
