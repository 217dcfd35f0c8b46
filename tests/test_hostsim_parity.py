"""Checks the arithmetic of the GPU row programs WITHOUT a GPU: tests/hostsim compiles
abr_control_amd/csrc/abrk_rows.h (exactly what one GPU lane executes) for the host.
Compared against the reference-generated golden vectors and the oracle, for the
compile-time specialised arms ("static") and the runtime-table arms ("rt").
The `-m gpu` tests repeat the same checks through libabrk.so on the device."""
import numpy as np
import pytest

from abr_control_amd import _abi
from tests import cases
from tests.conftest import golden

ARMS = ["twojoint", "threejoint", "ur5", "jaco2"]
DYN_ARMS = ARMS + ["onejoint"]  # N_LINKS = 1: kinematics of every frame, M = g = C = 0
BIG = {"ur5:cfg2": 1024, "ur5:cfg4": 512, "jaco2:cfg3": 512, "threejoint:cfg5": 1024}


@pytest.mark.parametrize("variant", ["static", "rt"])
@pytest.mark.parametrize("arm", DYN_ARMS)
def test_rows_dynamics_match_reference(arm, variant):
    cases.check_dynamics_against_golden(cases.HostsimBackend(arm, variant), arm, golden(arm))


@pytest.mark.parametrize("case_id", sorted(cases.CASES))
def test_rows_controllers_match_reference(case_id):
    arm = cases.CASES[case_id]["arm"]
    cases.check_case_against_golden(cases.HostsimBackend(arm, "static"), case_id, golden(arm), rows=BIG.get(case_id))


@pytest.mark.parametrize("case_id", ["twojoint:cfg1", "ur5:cfg2", "ur5:osc6_alg0", "ur5:osc_null2", "ur5:sliding",
                                     "jaco2:cfg3", "jaco2:osc6_alg1", "threejoint:cfg5", "ur5:joint"])
def test_rows_runtime_table_arms(case_id):
    arm = cases.CASES[case_id]["arm"]
    cases.check_case_against_golden(cases.HostsimBackend(arm, "rt"), case_id, golden(arm), rows=256)


def test_rows_fp32_config5():
    """BASELINE config 5: threejoint Sliding in fp32, tolerance 1e-4 vs the as-shipped reference"""
    g = golden("threejoint")
    be = cases.HostsimBackend("threejoint")
    u, _ = cases.run_case(be, cases.CASES["threejoint:cfg5"], g, dtype=np.float32, rows=1024)
    assert u.dtype == np.float32
    r = cases.rel_err(u.astype(float), g["cfg5_uS"][:1024])
    # J[:3] of the planar arm loses rank when the arm is (nearly) straight: pinv amplifies fp32 rounding there
    q = g["cfg5_q"][:1024]
    well = (np.abs(np.sin(q[:, 1])) > 0.05) & (np.abs(np.sin(q[:, 2])) > 0.05)
    assert r[well].max() <= cases.TOL_F32, r[well].max()


def test_rows_fp32_other_kernels():
    g = golden("ur5")
    be = cases.HostsimBackend("ur5")
    cases.check_case_against_golden(be, "ur5:cfg2", g, dtype=np.float32, rows=512)
    cases.check_case_against_golden(be, "ur5:joint", g, dtype=np.float32)


@pytest.mark.parametrize("case_id", ["twojoint:cfg1", "ur5:cfg4", "jaco2:cfg3", "ur5:osc6_alg0", "jaco2:osc6_alg1",
                                     "ur5:osc_null2", "ur5:sliding", "jaco2:damping"])
def test_rows_fp32_row_programs_match_reference(case_id):
    """the float instantiations of the row programs (the -m gpu suite repeats this on the device, where sin / cos come
    from the LDS table: test_gpu_fp32_kernels_match_reference)"""
    arm = cases.CASES[case_id]["arm"]
    cases.check_case_against_golden(cases.HostsimBackend(arm, "static"), case_id, golden(arm), dtype=np.float32)


@pytest.mark.parametrize("arm", ["ur5", "jaco2", "threejoint"])
def test_rows_fp32_dynamics_match_reference(arm):
    cases.check_dynamics_against_golden(cases.HostsimBackend(arm, "static"), arm, golden(arm), dtype=np.float32)


SIX_ROW_CASES = ["ur5:osc6_alg0", "ur5:osc6_alg1", "ur5:osc6_vmax", "ur5:osc_abg", "ur5:osc_xz_b", "ur5:osc_link5",
                 "jaco2:osc5", "jaco2:osc6_alg1", "threejoint:osc_xyg_alg0", "threejoint:osc_xyg_alg1"]


@pytest.mark.parametrize("case_id", SIX_ROW_CASES)
def test_rows_six_row_handover_form_matches_reference(case_id):
    """the two-pass form of the six-row law (first pass without the eigen-decomposition; a deferring row leaves Mx_inv,
    its task Jacobian rows, u_task and the joint-space sums in a record, osc6_finish_row completes it from there) against
    the reference's outputs - every truncating golden row goes through the record"""
    arm = cases.CASES[case_id]["arm"]
    g = golden(arm)
    be = cases.HostsimBackend(arm, "static", handover=True)
    r = cases.check_case_against_golden(be, case_id, g)
    assert be.deferred >= r["n_trunc_compared"], (be.deferred, r)


NOTS_CASES = [c for c in sorted(cases.CASES) if cases.takes_plain_six_row_law(c)]


def test_nots_case_list_covers_the_six_row_cases():
    """every golden case of the plain six-row law is run without the training signal too (the list is derived from the
    dispatch rule, abrk_params.h osc_fast_rows + Launch::osc_launch_feat, not kept by hand)"""
    assert set(NOTS_CASES) == set(SIX_ROW_CASES)


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("variant", ["static", "rt"])
@pytest.mark.parametrize("handover", [False, True], ids=["onepass", "handover"])
@pytest.mark.parametrize("case_id", NOTS_CASES)
def test_rows_six_row_cases_without_training_signal(case_id, handover, variant, dtype):
    """what bench.py times on the six-row law and what a C-ABI caller without a training-signal buffer runs: the NoTs
    arithmetic (gravity folded into the velocity term ahead of the factorisations), one-pass and hand-over forms, built-in
    and runtime-table row programs, both arithmetic types - against the reference's outputs"""
    arm = cases.CASES[case_id]["arm"]
    be = cases.HostsimBackend(arm, variant, handover=handover, training_signal=False)
    r = cases.check_case_against_golden(be, case_id, golden(arm), dtype=dtype)
    if handover and dtype == np.float64:
        assert be.deferred >= r["n_trunc_compared"], (be.deferred, r)


@pytest.mark.parametrize("arm", ["ur5", "jaco2"])
def test_rows_six_row_near_singular_postures_without_training_signal(arm):
    """the truncating rows of the near-singular postures through the NoTs row programs (plain law; the other two laws of
    the set carry optional inputs and have no NoTs twin), hand-over and one-pass forms"""
    be = cases.HostsimBackend(arm, "static", handover=True, training_signal=False)
    ref = cases.HostsimBackend(arm, "static", training_signal=False)
    worst, n_trunc = cases.check_six_row_near_singular(be, arm, B=300, reference=ref)
    assert be.deferred >= n_trunc and worst <= cases.TOL_D


def test_rows_fuzz_plain_six_row_law_without_training_signal():
    """random 1..7-joint user arms, any mask / frame / offset / vmax / orientation algorithm, no optional input and no
    training signal: the NoTs row programs against the oracle, one-pass and hand-over forms"""
    for fc in cases.fuzz_osc_cases(61, 16, plain_six=True):
        for handover in (False, True):
            cases.check_fuzz_case(lambda tab, h=handover: cases.HostsimBackend(tab, handover=h), fc, B=64)


@pytest.mark.parametrize("arm", ["ur5", "jaco2"])
def test_rows_six_row_handover_form_near_singular_postures(arm):
    """hundreds of truncating rows (postures next to the kinematic singularities) through the hand-over records: plain
    law, Coriolis + two fused secondary controllers, target velocity + integral state + external null-space signal -
    against the oracle and against the one-pass form of the same row programs"""
    be = cases.HostsimBackend(arm, "static", handover=True)
    worst, n_trunc = cases.check_six_row_near_singular(be, arm, B=300, reference=cases.HostsimBackend(arm, "static"))
    assert be.deferred >= n_trunc
    assert worst <= cases.TOL_D


@pytest.mark.parametrize("arm", ["ur5", "jaco2", "threejoint"])
def test_rows_six_row_handover_and_one_pass_forms_are_bit_equal(arm):
    """since round 5 every form of the six-row law hands a truncating row to ONE routine (abrk_ctrl.h osc6_tail) fed with
    the same values (osc6_jv / osc6_sums: contraction pinned off), from a record or from its registers: on the host
    build of the row programs the two-pass form and the one-pass form return the same bits - u and the training signal,
    random states and near-singular postures, plain law / Coriolis + secondary controllers / masked task rows, fp64 and
    fp32.  (The GPU suite asserts the same across the batch-size thresholds:
    G::test_gpu_six_row_bits_do_not_depend_on_the_batch_size.)"""
    n = _abi.load_table(arm)["n_joints"]
    rng = np.random.RandomState(17)
    q = rng.uniform(0, 2 * np.pi, (400, n))
    if arm != "threejoint":
        qs = cases.near_singular_postures(arm, 120)
        q[:len(qs)] = qs
    dq, t = rng.uniform(0, 5, (400, n)), rng.uniform(-1, 1, (400, 6))
    one, two = cases.HostsimBackend(arm, "static"), cases.HostsimBackend(arm, "static", handover=True)
    laws = [dict(kp=200, ko=150, kv=25, ctrlr_dof=[1] * 6),
            dict(kp=100, ko=60, kv=12, ctrlr_dof=[1, 1, 1, 1, 1, 0], use_C=True,
                 null_controllers=[_abi.make_damping(5), _abi.make_resting([0.3] * n, kp=20, kv=4)]),
            dict(kp=80, ko=40, ctrlr_dof=[1, 0, 1, 1, 0, 1], orientation_algorithm=1)]
    for kw in laws:
        p = _abi.make_osc_params(n, **kw)
        for dt in (np.float64, np.float32):
            u1, ts1 = one.osc(p, q, dq, t, dtype=dt)
            u2, ts2 = two.osc(p, q, dq, t, dtype=dt)
            assert np.array_equal(u1, u2, equal_nan=True) and np.array_equal(ts1, ts2, equal_nan=True), (kw, dt)
            assert np.isfinite(u1).all()
    assert two.deferred > 0, "no row went through the hand-over records"


def test_rows_six_row_handover_form_fuzz_on_user_arms():
    """random 1..7-joint user arms (runtime-table row programs), any ctrlr_dof mask / frame / optional input: the
    two-pass form against the oracle"""
    ran = 0
    for fc in cases.fuzz_osc_cases(41, 14):
        be = []

        def factory(tab):
            be.append(cases.HostsimBackend(tab, handover=True))
            return be[-1]

        cases.check_fuzz_case(factory, fc, B=64)
        ran += be[-1].deferred
    assert ran > 0, "no fuzz row went through the hand-over records"


@pytest.mark.parametrize("arm", ["ur5", "threejoint"])
def test_rows_obstacles_split_program_equals_one_pass(arm):
    """AvoidObstacles on orthogonal chains runs as phase A (kinematics, near tests, the cheap first two segments) + one
    heavy (obstacle, segment) pair at a time from the row's record + finish - on the GPU the pairs of a wavefront's 64
    rows are spread over its lanes through LDS (obstacles_lds_kernel).  Against the one-pass row program: the same
    contributions, summed in another order"""
    from tests import hostsim

    n = _abi.load_table(arm)["n_joints"]
    rng = np.random.RandomState(8)
    q = rng.uniform(0, 2 * np.pi, (700, n))
    sets = [([[0.3, 0.2, 0.4, 0.1], [-0.2, 0.4, 0.3, 0.05], [0.1, -0.3, 0.6, 0.15]], 0.3),
            ([[0.25, 0.1, 0.5, 0.1]], 0.4),
            ((rng.uniform(-0.6, 0.6, (16, 4)) * [1, 1, 1, 0.2] + [0, 0, 0.4, 0.12]).tolist(), 0.3),
            ((rng.uniform(-0.6, 0.6, (16, 4)) * [1, 1, 1, 0.1] + [0, 0, 0.4, 0.06]).tolist(), 5.0)]  # every pair near
    for obstacles, thr in sets:
        P = _abi.make_obstacles_params(obstacles, thr, 30)
        for dt, tol in ((np.float64, 1e-11), (np.float32, 2e-3)):
            a = hostsim.avoid_obstacles_generate(arm, P, q, dtype=dt).astype(float)
            b = hostsim.avoid_obstacles_generate(arm, P, q, dtype=dt, plain=True).astype(float)
            assert np.all(np.isfinite(a))
            # (clipping at +-maximum hides nothing here: both sides clip the same sums)
            assert np.max(np.abs(a - b)) <= tol * max(1.0, np.max(np.abs(b))), (arm, len(obstacles), thr, dt)


def test_rows_twojoint_closed_forms():
    """reference's analytic known answers (arms/tests/dummy_base_arm.py) on its test grids"""
    k = golden("known_answers")
    be = cases.HostsimBackend("twojoint")
    Q = k["q_grid"]
    for f in ("link0", "joint0", "link1", "joint1", "link2", "EE"):
        r = be.dynamics(Q, None, f, None, ("Tx", "J", "R", "Tinv"))
        assert np.allclose(r["Tx"], k[f"Tx_{f}"])
        assert np.allclose(r["J"], k[f"J_{f}"])
        assert np.allclose(r["R"], k[f"R_{f}"])
        assert np.allclose(r["Tinv"], k[f"Tinv_{f}"])
    r = be.dynamics(Q, None, "EE", None, ("M", "g"))
    assert np.allclose(r["M"], k["M"]) and np.allclose(r["g"], k["g"])
    QD = k["qdq_grid"]
    for f in ("link0", "joint0", "link1", "joint1", "link2", "EE"):
        r = be.dynamics(QD[:, :2], QD[:, 2:], f, None, ("dJ", "C"))
        assert np.allclose(r["dJ"], k[f"dJ_{f}"])
    assert np.allclose(r["C"], k["C"])


def test_rows_external_null_signal_equals_fused():
    """u_null_ext path (caller-evaluated secondary controller) == fused Damping (osc.py:310-318)"""
    from abr_control_amd._abi import make_damping, make_osc_params as P

    g = golden("jaco2")
    be = cases.HostsimBackend("jaco2")
    q, dq, t = g["cfg3_q"][:128], g["cfg3_dq"][:128], g["cfg3_target"][:128]
    u_fused, _ = be.osc(P(6, kp=200, null_controllers=[make_damping(10)]), q, dq, t)
    une = be.joint(make_damping(10), False, q, dq)
    u_ext, _ = be.osc(P(6, kp=200), q, dq, t, une=une)
    assert np.max(cases.rel_err(u_ext, u_fused)) < 1e-9
    assert np.max(cases.rel_err(u_fused, g["cfg3_uD"][:128])) < 1e-6


def test_rows_edge_inputs():
    """angles far outside [0, 2pi), negative velocities, zero state, exact singular task"""
    from abr_control_amd._abi import make_osc_params as P

    be, orc = cases.HostsimBackend("ur5"), cases.OracleBackend("ur5")
    rng = np.random.RandomState(7)
    q = np.vstack([np.zeros(6), rng.uniform(-500, 500, (30, 6)), np.full(6, np.pi / 2)])
    dq = rng.uniform(-8, 8, q.shape)
    t = rng.uniform(-1, 1, (len(q), 6))
    for p in (P(6, kp=200), P(6, kp=50, ko=30, ctrlr_dof=[1] * 6, orientation_algorithm=1)):
        u, _ = be.osc(p, q, dq, t)
        uo, _ = orc.osc(p, q, dq, t)
        assert np.all(np.isfinite(u))
        assert np.max(cases.rel_err(u, uo)) < 1e-6
    # planar arm asked to control z as well: Mx_inv is exactly singular -> pinv truncation
    be3, or3 = cases.HostsimBackend("threejoint"), cases.OracleBackend("threejoint")
    q3, dq3, t3 = rng.uniform(0, 6, (32, 3)), rng.uniform(0, 5, (32, 3)), rng.uniform(-1, 1, (32, 6))
    p = P(3, kp=40, ctrlr_dof=[1, 1, 1, 0, 0, 0])
    u, _ = be3.osc(p, q3, dq3, t3)
    uo, _ = or3.osc(p, q3, dq3, t3)
    assert np.all(np.isfinite(u)) and np.max(cases.rel_err(u, uo)) < 1e-6


@pytest.mark.parametrize("case_id", ["ur5:cfg2", "ur5:cfg4", "ur5:osc6_alg0", "ur5:osc6_alg1", "ur5:osc_xz_b",
                                     "ur5:osc_xyz_tvel", "ur5:osc6_vmax", "ur5:osc_null2", "jaco2:osc5",
                                     "threejoint:osc_xyg_alg1", "twojoint:cfg1"])
def test_rows_law_only_path(case_id):
    """abrk_osc_law_batch's row program (control law on caller-supplied J, M, g, Cdq, xyz, R - the
    duck-typed robot_config boundary, e.g. MujocoConfig): dynamics from the oracle, law on the row code."""
    from tests import hostsim

    case = cases.CASES[case_id]
    arm, key = case["arm"], case["key"]
    g = golden(arm)
    orc = cases.OracleBackend(arm)
    n = orc.n
    rows = min(200, len(g[f"{key}_q"]))
    q, dq, t = g[f"{key}_q"][:rows], g[f"{key}_dq"][:rows], g[f"{key}_target"][:rows]
    tv = g[f"{key}_tvel"][:rows] if case["tv"] else None
    params = case["params"](n)
    frame = "EE"
    d = orc.dynamics(q, dq, frame, None, ("J", "M", "g", "C", "Tx", "R"))
    Cdq = np.einsum("bij,bj->bi", d["C"], dq)
    une = None
    if params.n_null:  # secondary controllers are the caller's business on this path
        une = sum(orc.joint(params.null_ctrl[c], False, q, dq) for c in range(params.n_null))
        params.n_null = 0
    u, ts = hostsim.osc_law(n, params, d["J"], d["M"], dq, t, g=d["g"], Cdq=Cdq, xyz=d["Tx"], R=d["R"], q=q,
                            target_velocity=tv, u_null_ext=une)
    ok = np.ones(rows, bool)
    band = cases.threshold_band(g, key)
    if band is not None:
        ok &= ~band[:rows]
    tol = cases.TOL_THREEJOINT if arm == "threejoint" else 1e-6
    assert cases.rel_err(u, g[f"{key}_uD"][:rows])[ok].max() <= tol


@pytest.mark.parametrize("variant", ["static", "rt"])
def test_rows_closed_loop_rollout(variant):
    """the fused rollout row program (OSC.generate + ArmSim._step per step, state in registers)"""
    from tests import hostsim
    from tests.test_oracle_golden import _rollout_setup

    g, tab, plant, params = _rollout_setup()
    T, every = int(g["rollout_T"]), int(g["rollout_every"])
    arm = "twojoint" if variant == "static" else tab
    q, dq, qt, dqt, ut = hostsim.rollout_twolink(arm, params, plant, g["rollout_q0"], g["rollout_dq0"],
                                                 g["rollout_target"], T, every)
    assert np.max(np.abs(qt - g["rollout_qD"])) < 1e-9
    assert np.max(np.abs(dqt - g["rollout_dqD"])) < 1e-7
    assert np.max(np.abs(ut - g["rollout_uD"])) / np.max(np.abs(g["rollout_uD"])) < 1e-8
    assert np.array_equal(q, qt[:, -1])


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 6, 7])
def test_rows_user_arms_all_joint_counts(n):
    """runtime-table kernels for every supported joint count (1..ABRK_MAX_JOINTS) vs the oracle on
    synthetic arms, orthogonal and Jaco2-like non-orthogonal statics"""
    from abr_control_amd._abi import make_damping, make_joint, make_osc_params as P, make_sliding_params
    from oracle.oracle import Oracle
    from tests import hostsim
    from tests.synthetic_arms import make_arm

    for nonorth in (False, True):
        tab = make_arm(n, 100 + n, nonorth)
        o = Oracle(tab)
        if n == 5:  # N_LINKS short of the last link (the reference's jaco2 / onejoint quirk): M, g, C only
            tq = make_arm(n, 100 + n, nonorth, n_links_dyn=4)
            oq = Oracle(tq)
            rngq = np.random.RandomState(9)
            qq, dqq = rngq.uniform(-3, 3, (8, n)), rngq.uniform(-2, 2, (8, n))
            rq = hostsim.dynamics(tq, qq, dqq, None, None, ("M", "g", "C"))
            for b in range(8):
                assert np.allclose(rq["M"][b], oq.M(qq[b]), atol=1e-12)
                assert np.allclose(rq["g"][b], oq.g(qq[b]), atol=1e-12)
                assert np.allclose(rq["C"][b], oq.C(qq[b], dqq[b]), atol=1e-11)
        rng = np.random.RandomState(n)
        B = 24
        q, dq = rng.uniform(-3, 3, (B, n)), rng.uniform(-2, 2, (B, n))
        for frame in ("EE", f"link{n}", f"joint{n - 1}", "link0"):
            fid = _abi_frame(frame, n)
            r = hostsim.dynamics(tab, q, dq, fid, [0.05, -0.02, 0.03], ("Tx", "J", "dJ", "R", "T", "Tinv", "quat"))
            for b in range(B):
                assert np.allclose(r["Tx"][b], o.Tx(frame, q[b], [0.05, -0.02, 0.03]), atol=1e-12)
                assert np.allclose(r["J"][b], o.J(frame, q[b], [0.05, -0.02, 0.03]), atol=1e-12)
                assert np.allclose(r["dJ"][b], o.dJ(frame, q[b], dq[b], [0.05, -0.02, 0.03]), atol=1e-11)
                assert np.allclose(r["R"][b], o.R(frame, q[b]), atol=1e-13)
                assert np.allclose(r["Tinv"][b], o.T_inv(frame, q[b]), atol=1e-12)
        r = hostsim.dynamics(tab, q, dq, None, None, ("M", "g", "C"))
        for b in range(B):
            assert np.allclose(r["M"][b], o.M(q[b]), atol=1e-12)
            assert np.allclose(r["g"][b], o.g(q[b]), atol=1e-12)
            assert np.allclose(r["C"][b], o.C(q[b], dq[b]), atol=1e-11)
        t = rng.uniform(-0.5, 0.5, (B, 6))
        k = min(n, 3)
        dof = [1] * k + [0] * (6 - k)
        for p in (P(n, kp=30, ctrlr_dof=dof), P(n, kp=30, ctrlr_dof=dof, use_C=True, null_controllers=[make_damping(3)])):
            u = hostsim.osc_generate(tab, p, q, dq, t)
            uo = o.osc_batch(p, q, dq, t)
            cond_ok = np.array([np.linalg.cond(o.M(q[b])) < 1e8 for b in range(B)])
            assert cases.rel_err(u, uo)[cond_ok].max() < 1e-6
        if n >= 6:
            p = P(n, kp=30, ko=20, ctrlr_dof=[1] * 6, orientation_algorithm=1)
            assert cases.rel_err(hostsim.osc_generate(tab, p, q, dq, t), o.osc_batch(p, q, dq, t)).max() < 1e-6
        u = hostsim.joint_generate(tab, make_joint(10, 3), True, q, dq, q * 0.5)
        assert cases.rel_err(u, o.joint_batch(make_joint(10, 3), True, q, dq, q * 0.5)).max() < 1e-9
        if n >= 3:
            ps = make_sliding_params(n)
            u = hostsim.sliding_generate(tab, ps, q, dq, t[:, :3])
            uo, _ = o.sliding_batch(ps, q, dq, t[:, :3])
            assert cases.rel_err(u, uo).max() < 1e-6


def _abi_frame(name, n):

    return _abi.frame_id(name, n)


@pytest.mark.parametrize("variant", ["static", "rt"])
@pytest.mark.parametrize("arm", ["ur5", "jaco2"])
def test_rows_inverse_kinematics(arm, variant):
    from tests import hostsim

    g = golden(arm)
    a = arm if variant == "static" else _abi.load_table(arm)
    for method in (1, 2, 3):
        pp, vp = hostsim.ik_generate_path(a, _abi.make_ik_params(method=method), g["ik_q0"], g["ik_target"])
        assert np.max(np.abs(pp - g[f"ik_m{method}_posD"])) < 1e-9
        assert np.max(np.abs(vp - g[f"ik_m{method}_velD"])) < 1e-9
    # fewer joints than task dimensions (pinv of a 6 x 3 Jacobian), synthetic arm vs oracle
    from oracle import oracle as O
    from tests.synthetic_arms import make_arm

    tab = make_arm(3, 77)
    rng = np.random.RandomState(5)
    q0, tgt = rng.uniform(-1, 1, (4, 3)), rng.uniform(-0.4, 0.4, (4, 6))
    for method in (1, 2, 3):
        p = _abi.make_ik_params(method=method, n_timesteps=60)
        pp, _ = hostsim.ik_generate_path(tab, p, q0, tgt)
        po, _ = O.ik_paths(tab, p, q0, tgt)
        assert np.max(np.abs(pp - po)) < 1e-8


@pytest.mark.parametrize("variant", ["static", "rt"])
@pytest.mark.parametrize("arm", ARMS)
def test_rows_secondary_controllers(arm, variant):
    """AvoidJointLimits / Floating / AvoidObstacles row programs vs the reference's outputs"""
    rep = cases.check_secondary_against_golden(cases.HostsimBackend(arm, variant), arm, golden(f"sec_{arm}"))
    assert rep["obstacles_band"] <= 16  # of 128: pinv-threshold and noise-inversion rows
    cases.check_oscsec_against_golden(cases.HostsimBackend(arm, variant), arm, golden(f"sec_{arm}"))


def test_rows_secondary_fp32_and_accumulate():
    from tests import hostsim

    g = golden("sec_ur5")
    be = cases.HostsimBackend("ur5")
    cases.check_secondary_against_golden(be, "ur5", g, dtype=np.float32)
    # accumulate: u += signal
    P = cases.secondary_obstacle_params(g)
    q = g["obs_q"]
    base = np.full((len(q), 6), 0.25)
    u = hostsim.avoid_obstacles_generate("ur5", P, q, u=base.copy())
    assert np.allclose(u - base, hostsim.avoid_obstacles_generate("ur5", P, q), rtol=0, atol=1e-9)


@pytest.mark.parametrize("idx", range(24))
def test_rows_fuzz_osc_parameter_space(idx):
    """seeded random controller configurations on random user arms (runtime-table row programs) vs the oracle"""
    fc = cases.fuzz_osc_cases(7, 24)[idx]
    cases.check_fuzz_case(cases.HostsimBackend, fc)


@pytest.mark.parametrize("arm", ["twojoint", "threejoint"])
def test_rows_xy_fast_kernel_equals_general_kernel(arm):
    """ctrlr_dof = x,y on arms of <= 3 joints runs a two-row kernel; a zero external null signal forces the
    masked six-row kernel on the same inputs (target z != 0 enters the vmax norm, osc.py:198-215, in both)"""
    be = cases.HostsimBackend(arm)
    n = be.n
    rng = np.random.RandomState(3)
    B = 200
    q, dq, t = rng.uniform(0, 6.28, (B, n)), rng.uniform(-3, 3, (B, n)), rng.uniform(-1, 1, (B, 6))
    for kw in (dict(kp=20, kv=5), dict(kp=20, kv=5, vmax=[0.5, 0.5], use_C=True, xyz_offset=[0.1, -0.05, 0.0]),
               dict(kp=30, ki=0.2, null_controllers=[cases.make_damping(4)])):
        p = cases.P(n, ctrlr_dof=cases.XY, **kw)
        ie1 = np.zeros((B, 6)) if kw.get("ki") else None
        ie2 = np.zeros((B, 6)) if kw.get("ki") else None
        u1, ts1 = be.osc(p, q, dq, t, ie=ie1)
        u2, ts2 = be.osc(p, q, dq, t, ie=ie2, une=np.zeros((B, n)))
        assert np.allclose(u1, u2, rtol=1e-11, atol=1e-11) and np.allclose(ts1, ts2, rtol=1e-11, atol=1e-11)
        if ie1 is not None:
            assert np.allclose(ie1, ie2, rtol=0, atol=1e-14)


@pytest.mark.parametrize("arm", ["ur5", "threejoint"])
def test_rows_six_row_use_C_on_orthogonal_chains(arm):
    assert cases.check_six_row_use_C(cases.HostsimBackend(arm), arm) < 1e-6


@pytest.mark.parametrize("seed", range(12))
def test_rows_fuzz_sliding_joint_dynamics(seed):
    cases.check_fuzz_other(cases.HostsimBackend, seed)


@pytest.mark.parametrize("arm", ["ur5", "jaco2", "threejoint"])
def test_rows_osc_helper_methods_vs_reference(arm):
    """the row programs behind OSC._Mx / ._velocity_limiting / ._calc_orientation_forces against the reference's own
    outputs (tests/golden/oschelpers_<arm>.npz; inputs and expectations of controllers/tests/test_osc.py)"""
    from tests import hostsim

    g = golden(f"oschelpers_{arm}")
    kp, ko, kv, v0, v1 = g["vl_gains"]
    n = g["mx_q"].shape[1]
    p = _abi.make_osc_params(n, kp=kp, ko=ko, kv=kv, vmax=[v0, v1], ctrlr_dof=[1] * 6)
    assert np.max(np.abs(hostsim.osc_velocity_limiting(p, g["vl_in"]) - g["vl_out"])) < 1e-12
    for alg in (0, 1):
        assert np.max(np.abs(hostsim.osc_orientation_forces(alg, g["of_R"], g["of_abg"]) - g[f"of_alg{alg}"])) < 1e-9
    M = g["mx_M"]
    B = len(M)
    nrm = lambda a: np.maximum(np.abs(a).max(axis=(1, 2)), 1e-300)
    Mx, Minv = hostsim.osc_mx(n, M, np.broadcast_to(np.eye(n), (B, n, n)).copy(), 1e-5)
    assert np.allclose(Mx, M, atol=1e-5)  # test_osc.py:80
    assert (np.abs(Mx - g["mx_eye_Mx"]).max(axis=(1, 2)) / nrm(g["mx_eye_Mx"])).max() < 1e-9
    assert (np.abs(Minv - g["mx_eye_Minv"]).max(axis=(1, 2)) / nrm(g["mx_eye_Minv"])).max() < 1e-9
    Mx1, _ = hostsim.osc_mx(n, M, np.ones((B, 6, n)))
    assert all(np.all(np.linalg.svd(x)[1][1:] < 1e-10) for x in Mx1)  # test_osc.py:86
    assert (np.abs(Mx1 - g["mx_ones_Mx"]).max(axis=(1, 2)) / nrm(g["mx_ones_Mx"])).max() < 1e-9
    for k in (1, 2, 3, 6):
        if f"mx_k{k}_J" not in g.files:
            continue
        ok = cases.mx_rows_clear_of_thresholds(g[f"mx_k{k}_det"], g[f"mx_k{k}_sv"])
        Mxk, _ = hostsim.osc_mx(n, M, g[f"mx_k{k}_J"])
        err = np.abs(Mxk - g[f"mx_k{k}_Mx"]).max(axis=(1, 2)) / nrm(g[f"mx_k{k}_Mx"])
        assert err[ok].max() <= 1e-7, (k, err[ok].max())


@pytest.mark.parametrize("seed", range(20, 32))
def test_rows_fuzz_secondary_controllers(seed):
    """AvoidJointLimits / Floating / AvoidObstacles on random user arms with random parameters vs the oracle"""
    cases.check_fuzz_secondary(cases.HostsimBackend, seed)


def test_rows_truncated_pinv_rows_are_compared():
    """every golden row on which `pinv(Mx_inv, rcond=1e-4)` really truncates (osc.py:142-145; 49 rows over the OSC
    cases) takes part in the golden assert - only rows within 1e-6 (relative) of a threshold may be excluded"""
    tot = cmp_ = 0
    for case_id, case in sorted(cases.CASES.items()):
        if case["kind"] != "osc":
            continue
        r = cases.check_case_against_golden(cases.HostsimBackend(case["arm"], "static"), case_id, golden(case["arm"]))
        tot += r["n_trunc"]
        cmp_ += r["n_trunc_compared"]
        assert r["n_band"] <= 1, (case_id, r)
    assert tot >= 45 and cmp_ == tot, (tot, cmp_)


@pytest.mark.parametrize("arm", ["ur5", "jaco2"])
def test_rows_quaternion_every_frame(arm):
    """power-iteration quaternion vs the reference's eigh on every frame, incl. Jaco2's non-orthogonal late frames"""
    cases.check_quaternions_all_frames(cases.HostsimBackend(arm, "static"), arm, golden(f"quat_{arm}"))


@pytest.mark.parametrize("arm,kw", [
    ("ur5", dict(kp=200)),
    ("ur5", dict(kp=100, ko=60, kv=12, ctrlr_dof=[1] * 6, use_C=True, ref_frame="link5", xyz_offset=[0.05, 0.0, -0.1],
                 null_controllers=[cases.make_damping(5)])),
    ("jaco2", dict(kp=200, null_controllers=[cases.make_damping(10)])),
    ("twojoint", dict(kp=10, kv=3, ctrlr_dof=cases.XY, use_C=True)),
])
def test_rows_fused_full_outputs(arm, kw):
    """the Mode-F row program (osc_full_body: u + Tx, J, M, g) on the host: u / training signal equal to the plain row
    program, the robot_config outputs equal to the dynamics row program of the same frame / offset, static and
    runtime-table arms"""
    from tests import hostsim

    tab = _abi.load_table(arm)
    n = tab["n_joints"]
    p = cases.P(n, **kw)
    rng = np.random.RandomState(5)
    q, dq, t = rng.uniform(0, 2 * np.pi, (40, n)), rng.uniform(0, 5, (40, n)), rng.uniform(-1, 1, (40, 6))
    frame, off = kw.get("ref_frame", "EE"), kw.get("xyz_offset")
    for a in (arm, tab):
        u0, ts0 = hostsim.osc_generate(a, p, q, dq, t, training_signal=True)
        u1, ts1, dyn = hostsim.osc_generate_full(a, p, q, dq, t)
        ref = hostsim.dynamics(a, q, None, _abi.frame_id(frame, n), off, ("Tx", "J", "M", "g"), np.float64)
        assert np.allclose(u1, u0, rtol=1e-12, atol=1e-12) and np.allclose(ts1, ts0, rtol=1e-12, atol=1e-12)
        for k in ("Tx", "J", "M", "g"):
            assert np.allclose(dyn[k], ref[k], rtol=1e-13, atol=1e-13), (arm, k)
        # with the velocity-dependent outputs (C, dJ): the variant whose dynamics pass assembles the Christoffel matrix
        # and whose Coriolis vector is C dq from it
        allw = ("Tx", "J", "M", "g", "C", "dJ")
        u2, ts2, dyn2 = hostsim.osc_generate_full(a, p, q, dq, t, want=allw)
        ref2 = hostsim.dynamics(a, q, dq, _abi.frame_id(frame, n), off, allw, np.float64)
        scale = max(1.0, float(np.max(np.abs(u0))))
        assert np.max(np.abs(u2 - u0)) <= 1e-11 * scale and np.max(np.abs(ts2 - ts0)) <= 1e-11 * scale
        for k in allw:
            assert np.allclose(dyn2[k], ref2[k], rtol=1e-12, atol=1e-12), (arm, k)
        only = hostsim.osc_generate_full(a, p, q, dq, t, want=("C",))[2]
        assert list(only) == ["C"] and np.allclose(only["C"], ref2["C"], rtol=1e-12, atol=1e-12)


def test_rows_six_row_law_without_training_signal():
    """the NoTs variant of the six-row law (no training signal asked for: the gravity term joins the velocity term before
    the factorisations, which is what lets the GPU's first-pass kernel keep its law free of spills) against the plain
    variant: the same u to rounding, on the UR5 (orthogonal chain, with and without the Coriolis term) and Jaco2"""
    from tests import hostsim

    rng = np.random.RandomState(11)
    for arm, kw in (("ur5", dict(kp=100, ko=60, kv=12, ctrlr_dof=[1] * 6)),
                    ("ur5", dict(kp=100, ko=60, kv=12, ctrlr_dof=[1] * 6, use_C=True, orientation_algorithm=1)),
                    ("ur5", dict(kp=50, ctrlr_dof=[1, 1, 0, 1, 0, 1], use_g=False)),
                    ("jaco2", dict(kp=80, ko=40, ctrlr_dof=[1] * 6, use_C=True))):
        p = cases.P(6, **kw)
        q, dq, t = rng.uniform(0, 2 * np.pi, (300, 6)), rng.uniform(0, 5, (300, 6)), rng.uniform(-1, 1, (300, 6))
        u_ts, _ = hostsim.osc_generate(arm, p, q, dq, t, training_signal=True)
        u_no = hostsim.osc_generate(arm, p, q, dq, t, training_signal=False)
        scale = np.max(np.abs(u_ts), axis=1, keepdims=True)
        assert np.max(np.abs(u_no - u_ts) / scale) < 1e-13, (arm, kw)


def test_rows_six_by_six_eigensolvers():
    """abrk_ctrl.h `ql_eig` (Householder tridiagonalisation + implicit QL, every index a compile-time constant) - what
    the six-row OSC law's truncating pinv runs on since round 3 - and `jacobi_eig` (the cyclic Jacobi it replaced there,
    still used up to 3 x 3) against numpy.linalg.eigh / pinv(rcond=1e-4) on the matrices that break eigen-solvers:
    spectra graded over ten decades, multiples of the identity, rank 1 / 3 / 4, clustered pairs, one eigenvalue at
    rounding level, indefinite matrices, isolated zero rows (what the law passes for unselected task rows), diagonal
    and zero input."""
    from tests import hostsim

    rng = np.random.RandomState(3)
    mats = []
    for trial in range(6000):
        kind = trial % 10
        Q, _ = np.linalg.qr(rng.randn(6, 6))
        lam = {0: lambda: 10 ** rng.uniform(-8, 2, 6), 1: lambda: np.ones(6) * rng.uniform(.1, 10),
               2: lambda: np.r_[rng.uniform(.5, 2, 3), 0, 0, 0], 3: lambda: np.r_[rng.uniform(.5, 2, 1), np.zeros(5)],
               4: lambda: np.r_[1.0, 1.0 + 1e-9, rng.uniform(0.1, 1, 4)],
               5: lambda: np.r_[rng.uniform(1, 2, 5), 10 ** rng.uniform(-16, -6)], 6: lambda: rng.uniform(-1, 1, 6),
               7: lambda: 10 ** rng.uniform(-3, 3, 6), 8: lambda: np.r_[10 ** rng.uniform(-2, 2, 4), 0, 0],
               9: lambda: 10 ** rng.uniform(-5, 0, 6)}[kind]()
        A = (Q * lam) @ Q.T
        if kind == 8:
            B4 = rng.randn(4, 4)
            A = np.zeros((6, 6))
            A[np.ix_([0, 2, 3, 5], [0, 2, 3, 5])] = B4 @ B4.T
        if kind == 1 and trial % 20 == 1:
            A = np.diag(10 ** rng.uniform(-3, 3, 6))
        if kind == 3 and trial % 20 == 3:
            A = np.zeros((6, 6))
        mats.append((A + A.T) / 2)
    A = np.array(mats)
    ref = np.linalg.eigvalsh(A)
    nrm = np.maximum(np.abs(ref).max(axis=1), 1e-300)
    ratio = np.abs(ref) / nrm[:, None]
    off_band = ~(np.abs(ratio - 1e-4) < 1e-9).any(axis=1)
    Pn = np.array([np.linalg.pinv(a, rcond=1e-4, hermitian=True) for a in A])
    for method in (0, 1):
        lam, V = hostsim.sym6_eig(A, method)
        assert np.isfinite(lam).all() and np.isfinite(V).all()
        assert (np.abs(A @ V - V * lam[:, None, :]).max(axis=(1, 2)) / nrm).max() < 1e-14
        assert np.abs(np.einsum("bki,bkj->bij", V, V) - np.eye(6)).max() < 2e-14
        assert (np.abs(np.sort(lam, axis=1) - ref).max(axis=1) / nrm).max() < 1e-14
        # the truncated pseudo-inverse as osc_law6 / mx_row assemble it
        keep = np.abs(lam) > 1e-4 * np.abs(lam).max(axis=1, keepdims=True)
        w = np.where(keep, 1 / np.where(keep, lam, 1), 0)
        P = np.einsum("bai,bci,bi->bac", V, V, w)
        err = np.abs(P - Pn).max(axis=(1, 2)) / np.maximum(np.abs(Pn).max(axis=(1, 2)), 1e-300)
        assert err[off_band].max() < 1e-9, (method, err[off_band].max())


def test_rows_six_row_tail_early_exit_is_the_exact_pseudo_inverse():
    """abrk_ctrl.h `osc6_tail` (round 5): the truncating pseudo-inverse of the six-row law behind a QL iteration that STOPS
    once the still-coupled part of the tridiagonal matrix is certainly kept (Sturm pivots against a Gershgorin-bounded
    cut-off) and applies that part by a tridiagonal solve.  Against numpy.linalg.pinv(rcond=1e-4) on Mx_inv of truncating
    UR5 states, Jaco2's five-row setting (a masked LAST task row: an exact zero row / column), spectra graded over 14
    decades, eigenvalues just either side of the cut-off, clusters, rank deficiency, masked rows anywhere: exact to
    rounding - and the iteration really does stop early (UR5: after the first or second eigenvalue; Jaco2 five rows:
    mostly without isolating any)."""
    from oracle.oracle import Oracle
    from tests import hostsim

    rng = np.random.RandomState(11)
    mats, kinds = [], []
    for arm in ("ur5", "jaco2"):
        o = Oracle(_abi.load_table(arm))
        n = 0
        while n < 120:
            q = rng.uniform(0, 2 * np.pi, 6)
            J, M = o.J("EE", q), o.M(q)
            A = J @ np.linalg.inv(M) @ J.T
            if arm == "jaco2":
                A[5, :] = 0
                A[:, 5] = 0
            sv = np.linalg.eigvalsh(A)
            if abs(np.linalg.det(A)) < 1e-3 and sv[0] < 1e-4 * sv[-1]:
                mats.append(A)
                kinds.append(arm)
                n += 1
    for k in range(900):
        Q, _ = np.linalg.qr(rng.normal(size=(6, 6)))
        kind = k % 9
        if kind == 0:
            lam = 10.0 ** rng.uniform(-12, 2, 6)
        elif kind == 1:
            lam = np.array([1, 0.5, 0.2, 2e-4, 0.9e-4, 1e-9]) * 10 ** rng.uniform(-3, 3)
        elif kind == 2:
            lam = np.array([1, 1, 1, 1e-6, 1e-6, 1e-6])
        elif kind == 3:
            lam = np.array([3.0, 1e-5, 1e-5, 1e-5, 1e-5, 0.0])
        elif kind == 4:
            lam = np.concatenate([[1.0], 10.0 ** rng.uniform(-4.5, -3.5, 5)])
        elif kind == 5:
            lam = np.array([1, 0.3, 0.1, 0.03, 0.01, 10.0 ** rng.uniform(-9, -4.2)])
        else:
            lam = 10.0 ** rng.uniform(-1, 1, 6) if k % 2 else np.concatenate([10.0 ** rng.uniform(-1, 1, 5), [10.0 ** rng.uniform(-9, -5)]])
        A = (Q * lam) @ Q.T
        A = (A + A.T) / 2
        if kind >= 6:  # masked task rows: exact zero rows / columns
            for r in [(5,), (4, 5), (1, 5), (0,), (2, 3, 4, 5), (1,)][(k // 9) % 6]:
                A[r, :] = 0
                A[:, r] = 0
        mats.append(A)
        kinds.append("masked" if kind >= 6 else "synthetic")
    A = np.array(mats)
    kinds = np.array(kinds)
    n = len(A)
    G, b = rng.normal(size=(n, 8, 6)), rng.normal(size=(n, 12))
    u, ts, lexit, cut = hostsim.osc6_tail(A, G, b)
    assert np.isfinite(u).all() and np.isfinite(ts).all()
    worst = 0.0
    for k in range(n):
        sv = np.abs(np.linalg.eigvalsh(A[k]))
        if np.any(np.abs(sv / sv.max() - 1e-4) < 1e-6 * 1e-4):
            continue  # within 1e-6 of the cut-off: either answer is legitimate
        P = np.linalg.pinv(A[k], rcond=1e-4, hermitian=True)
        ts_ref = b[k, :6] - G[k, :6] @ (P @ G[k, 6])
        u_ref = ts_ref + b[k, 6:] - G[k, :6] @ (P @ G[k, 7])
        scale = np.abs(P).max() * np.abs(G[k]).max() ** 2 * 6
        worst = max(worst, np.abs(u[k] - u_ref).max() / scale, np.abs(ts[k] - ts_ref).max() / scale)
    assert worst < 1e-10, worst
    # the iteration stops early where it can
    ur5 = lexit[kinds == "ur5"]
    assert (ur5 <= 1).mean() > 0.95 and (ur5 == 0).mean() > 0.6, np.bincount(ur5 + 1)
    j2 = lexit[kinds == "jaco2"]
    assert (j2 <= 1).mean() > 0.95 and (j2 == -1).mean() > 0.5, np.bincount(j2 + 1)
    # ... and runs to the end where it must (eigenvalues all around the cut-off)
    assert (lexit[kinds == "synthetic"] == 5).any()


def test_rows_direct_sym3_eigensolver():
    """abrk_ctrl.h `sym3_eig` (the sweep-free eigen-decomposition behind AvoidObstacles' truncated pinv) against
    numpy.linalg.eigvalsh / pinv on the matrices that break closed forms: rank 1 and rank 2 (what the first two
    segments of every arm produce), clustered pairs at either end, spectra graded over 12 decades, multiples of the
    identity, diagonal input - eigenvalues to a few eps |A|, V orthonormal, pinv(rcond=0.01) as numpy's"""
    from tests import hostsim

    rng = np.random.RandomState(7)
    mats, kinds = [], []
    for trial in range(6000):
        kind = trial % 8
        Q, _ = np.linalg.qr(rng.randn(3, 3))
        if kind == 0:
            lam = np.array([1.0, 0, 0]) * rng.uniform(0.1, 10)
        elif kind == 1:
            lam = np.array([1.0, rng.uniform(1e-3, 1), 0])
        elif kind == 2:
            lam = 10 ** rng.uniform(-12, 0, 3)
        elif kind == 3:
            lam = np.array([1.0, 1 + rng.uniform(-1e-9, 1e-9), rng.uniform(0, 1e-3)])
        elif kind == 4:
            l1 = rng.uniform(0, 1e-3)
            lam = np.array([1.0, l1, l1 * (1 + rng.uniform(-1e-9, 1e-9))])
        elif kind == 5:
            lam = rng.uniform(0, 1, 3)
        elif kind == 6:
            lam = np.full(3, rng.uniform(0.1, 5)) * (1 + rng.uniform(-1e-15, 1e-15, 3))
        else:
            lam, Q = rng.uniform(0, 1, 3), np.eye(3)[rng.permutation(3)]
        A = (Q * lam) @ Q.T
        mats.append((A + A.T) / 2)
        kinds.append(kind)
    A = np.array(mats)
    lam, V = hostsim.sym3_eig(A)
    ref = np.linalg.eigvalsh(A)
    scale = np.abs(ref).max(axis=1)
    assert (np.abs(np.sort(lam, axis=1) - ref).max(axis=1) / scale).max() < 2e-14
    assert np.abs(np.einsum("bij,bik->bjk", V, V) - np.eye(3)).max() < 1e-14
    # A V = V diag(lam)
    assert (np.abs(np.einsum("bij,bjk->bik", A, V) - V * lam[:, None, :]).max(axis=(1, 2)) / scale).max() < 1e-13
    sv = np.abs(ref) / scale[:, None]
    clear = np.all(np.abs(sv - 0.01) > 1e-6, axis=1)  # away from the cut, where either answer is legitimate
    cut = 0.01 * np.abs(lam).max(axis=1, keepdims=True)
    w = np.where(np.abs(lam) > cut, 1.0 / np.where(lam == 0, 1, lam), 0.0)
    P = np.einsum("bij,bj,bkj->bik", V, w, V)
    Pr = np.array([np.linalg.pinv(a, rcond=0.01, hermitian=True) for a in A])
    err = np.abs(P - Pr).max(axis=(1, 2)) / np.abs(Pr).max(axis=(1, 2))
    assert err[clear].max() < 1e-11, (err[clear].max(), kinds[int(np.argmax(np.where(clear, err, 0)))])
    # fp32 instantiation: the same to single precision
    lam32, V32 = hostsim.sym3_eig(A, dtype=np.float32)
    assert (np.abs(np.sort(lam32, axis=1) - ref).max(axis=1) / scale).max() < 2e-5
    assert np.abs(np.einsum("bij,bik->bjk", V32, V32) - np.eye(3)).max() < 1e-5


def test_rows_cofactor_inverse_of_the_three_row_law():
    """`spd_inverse_small` (abrk_ctrl.h): the x,y,z / x,y law takes Mx = (J M^-1 J^T)^-1 (osc.py:135-141) by cofactors
    instead of a Cholesky factor.  Cofactors carry a relative error of eps * trace^K / det (eps * cond while one
    eigenvalue is small, eps * cond^2 when two are), which is why osc_law only keeps the result up to trace^K / det =
    1e6 and factorises beyond.  Against numpy.linalg.inv / det on graded SPD matrices: inside the gate the entries are
    good to 8 eps trace^K / det of |A^-1| (<= 2e-9) and det to the same; indefinite, singular and NaN matrices are
    refused (ok = False), which sends the row to the pinv branch."""
    from tests import hostsim as hs

    rng = np.random.RandomState(5)
    for K in (3, 2):
        mats = []
        for logc in (0, 1, 2, 3, 4, 6) if K == 3 else (0, 2, 4, 6, 8, 10):
            for _ in range(300):
                Q, _r = np.linalg.qr(rng.randn(K, K))
                lam = np.concatenate([[1.0], 10.0 ** (-logc * rng.uniform(0.3, 1.0, K - 1))]) * 10.0 ** rng.uniform(-3, 3)
                mats.append((Q * lam) @ Q.T)
        A = np.array([(m + m.T) / 2 for m in mats])
        inv, det, ok = hs.spd_inverse_small(A)
        assert ok.all()
        ref, dref = np.linalg.inv(A), np.linalg.det(A)
        amp = np.trace(A, axis1=1, axis2=2) ** K / dref  # the gate's quantity
        gate = amp <= 1e6
        assert gate.sum() > 600 and (~gate).sum() > 100  # both sides of the gate are exercised
        err = np.abs(inv - ref).max(axis=(1, 2)) / np.abs(ref).max(axis=(1, 2))
        bound = 8 * 2.3e-16 * amp + 1e-15
        assert (err[gate] <= bound[gate]).all(), (K, (err[gate] / bound[gate]).max())
        assert (np.abs(det / dref - 1)[gate] <= bound[gate]).all()
        assert err[gate].max() < 2e-9
        assert np.array_equal(inv, np.swapaxes(inv, 1, 2))  # symmetric by construction
        bad = np.array([np.diag([1.0, -1.0, 2.0][:K]), np.zeros((K, K)), np.full((K, K), np.nan),
                        np.ones((K, K)), -np.eye(K)])
        _i, _d, okb = hs.spd_inverse_small(bad)
        assert not okb.any()


def test_rows_three_row_law_near_singular_postures():
    """cases.check_near_singular_postures on the host build of the row programs: both sides of the cofactor / Cholesky
    gate of osc_law and the truncating pinv behind it are taken"""
    worst, beyond, trunc = cases.check_near_singular_postures(cases.HostsimBackend("ur5"))
    assert beyond > 100 and trunc > 100, (beyond, trunc)
    assert worst < 1e-6
