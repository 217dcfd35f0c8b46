"""Worker for gen_golden.py: one arm (or the known-answer grids) per process.

TEST INFRASTRUCTURE.  Imports the reference (scratch copy on PYTHONPATH) and records
its outputs; never imported by the package, never shipped to the GPU path.
"""
import importlib
import sys

import numpy as np

what, OUT = sys.argv[1], sys.argv[2]


# ----------------------------------------------------------------------------------------
class RawConfig:
    """robot_config whose wrappers skip the float32 cast = Oracle-D.

    Mirrors abr_control/arms/base_config.py:210-336 wrapper by wrapper, calling the very
    same generated functions (rc._M, rc._J[...], ...) but keeping their fp64 output.
    """

    def __init__(self, rc):
        self.rc = rc
        self.N_JOINTS = rc.N_JOINTS
        self.N_LINKS = rc.N_LINKS
        self.x_zeros = np.zeros(3)

    def _fn(self, name, x):
        x = self.x_zeros if x is None else x
        return x, (name + "[0,0,0]" if np.allclose(x, 0) else name)

    def g(self, q):
        self.rc.g(q)
        return np.array(self.rc._g(*tuple(q)), dtype="float64").flatten()

    def dJ(self, name, q, dq, x=None):
        self.rc.dJ(name, q, dq, x)
        x, fn = self._fn(name, x)
        return np.array(self.rc._dJ[fn](*(tuple(q) + tuple(dq) + tuple(x))), dtype="float64")

    def J(self, name, q, x=None):
        self.rc.J(name, q, x)
        x, fn = self._fn(name, x)
        return np.array(self.rc._J[fn](*(tuple(q) + tuple(x))), dtype="float64")

    def M(self, q):
        self.rc.M(q)
        return np.array(self.rc._M(*tuple(q)), dtype="float64")

    def R(self, name, q):
        self.rc.R(name, q)
        return np.array(self.rc._R[name](*tuple(q)), dtype="float64")

    def quaternion(self, name, q):
        from abr_control.utils import transformations

        R = self.R(name, q)
        return transformations.unit_vector(transformations.quaternion_from_matrix(matrix=R))

    def C(self, q, dq):
        self.rc.C(q, dq)
        return np.array(self.rc._C(*(tuple(q) + tuple(dq))), dtype="float64")

    def Tx(self, name, q, x=None):
        return self.rc.Tx(name, q, x)

    def T_inv(self, name, q, x=None):  # base_config.py:394-415 returns the generated fp64 matrix as is
        return self.rc.T_inv(name, q, x)


def frames_of(rc):
    fr = []
    for i in range(rc.N_JOINTS + 1):
        fr.append(f"link{i}")
        if i < rc.N_JOINTS:
            fr.append(f"joint{i}")
    fr.append("EE")
    return fr


def draw(rng, B, n, nt=6):
    """The reference benchmark's input distribution (examples/timing_plots.py:18-20)."""
    q = rng.uniform(0, 2 * np.pi, (B, n))
    dq = rng.uniform(0, 5, (B, n))
    tgt = rng.uniform(-1, 1, (B, nt))
    return q, dq, tgt


def mx_diag(raw, q, dof, ref_frame="EE", x=None):
    """det and singular values of Mx_inv (fp64) - lets tests find threshold-band states."""
    J = raw.J(ref_frame, q, x)[np.asarray(dof, dtype=bool)]
    M = raw.M(q)
    A = J @ np.linalg.inv(M) @ J.T
    s = np.linalg.svd(A, compute_uv=False)
    return np.linalg.det(A), s


# ----------------------------------------------------------------------------------------
def gen_arm(arm):
    mod = importlib.import_module(f"abr_control.arms.{arm}")
    from abr_control.controllers import OSC, Damping, Joint, RestingConfig, Sliding

    rc = mod.Config(use_cython=True)
    raw = RawConfig(rc)
    n = rc.N_JOINTS
    out = {}

    # ---------------- dynamics: raw fp64 outputs of the generated functions ------------
    rng = np.random.RandomState(0)
    Nd = 48
    q, dq, _ = draw(rng, Nd, n)
    q[0] = 0.0
    q[1] = np.asarray(rc.START_ANGLES, dtype=float)
    q[2] = -q[2]  # negative angles too
    q[3] = q[3] * 20.0 - 50.0  # well outside [0, 2pi)
    dq[4] = -dq[4]
    out["dyn_q"], out["dyn_dq"] = q, dq
    frames = frames_of(rc)
    out["frames"] = np.array(frames)
    xoff = np.array([0.11, -0.23, 0.37])
    out["xoff"] = xoff
    for f in frames:
        out[f"Tx_{f}"] = np.array([rc.Tx(f, q[i]) for i in range(Nd)])
        out[f"J_{f}"] = np.array([raw.J(f, q[i]) for i in range(Nd)])
    # every frame's rotation / full transform where cheap (small arms), EE for all
    rframes = frames if n <= 3 else ["EE"]
    for f in rframes:
        out[f"R_{f}"] = np.array([raw.R(f, q[i]) for i in range(Nd)])
        out[f"dJ_{f}"] = np.array([raw.dJ(f, q[i], dq[i]) for i in range(Nd)])
    out["T_EE"] = np.array([np.array(rc.T("EE", q[i]), dtype=float) for i in range(Nd)])
    out["M"] = np.array([raw.M(q[i]) for i in range(Nd)])
    out["g"] = np.array([raw.g(q[i]) for i in range(Nd)])
    out["C"] = np.array([raw.C(q[i], dq[i]) for i in range(Nd)])
    out["quat_EE"] = np.array([raw.quaternion("EE", q[i]) for i in range(Nd)])
    # point offset inside the EE frame (exercises the x != 0 generated variants)
    out["Tx_EE_x"] = np.array([rc.Tx("EE", q[i], x=xoff) for i in range(Nd)])
    out["J_EE_x"] = np.array([raw.J("EE", q[i], x=xoff) for i in range(Nd)])
    if n <= 3 or arm == "ur5":
        out["dJ_EE_x"] = np.array([raw.dJ("EE", q[i], dq[i], x=xoff) for i in range(Nd)])
    if n <= 3:
        out["Tinv_EE"] = np.array([np.array(rc.T_inv("EE", q[i]), dtype=float) for i in range(Nd)])
    print("  dynamics done; function types:", type(rc._M).__name__, type(rc._C).__name__, flush=True)

    # ---------------- controllers: Oracle-S and Oracle-D per case -----------------------
    xyz = [True, True, True, False, False, False]
    six = [True] * 6

    def run_osc(key, B, seed, kwargs, null=None, gen_kwargs=None, steps=1, tv=False, with_diag=True):
        """null: list of (cls, kwargs); builds separate controller objects for S and D."""
        rng = np.random.RandomState(seed)
        q, dq, tgt = draw(rng, B, n)
        tvel = rng.uniform(-0.5, 0.5, (B, 6)) if tv else None
        gen_kwargs = gen_kwargs or {}
        res = {}
        for label, cfg in (("S", rc), ("D", raw)):
            nulls = None
            if null:
                nulls = [cls(cfg, **kw) for cls, kw in null]
            us = np.zeros((steps, B, n))
            ts = np.zeros((steps, B, n))
            for b in range(B):
                ctrlr = OSC(cfg, null_controllers=nulls, **kwargs)  # fresh state per row
                for s in range(steps):
                    extra = dict(gen_kwargs)
                    if tvel is not None:
                        extra["target_velocity"] = tvel[b]
                    # multi-step cases: same (q,dq,target) each step -> only the
                    # integrated_error state evolves (osc.py:262-264)
                    us[s, b] = ctrlr.generate(q[b], dq[b], tgt[b], **extra)
                    ts[s, b] = ctrlr.training_signal
            res[label] = (us, ts)
        out[f"{key}_q"], out[f"{key}_dq"], out[f"{key}_target"] = q, dq, tgt
        if tvel is not None:
            out[f"{key}_tvel"] = tvel
        for label in ("S", "D"):
            us, ts = res[label]
            out[f"{key}_u{label}"] = us if steps > 1 else us[0]
            out[f"{key}_ts{label}"] = ts if steps > 1 else ts[0]
        if with_diag:
            dof = kwargs.get("ctrlr_dof", xyz)
            dets = np.zeros(B)
            svs = np.zeros((B, int(np.sum(dof))))
            for b in range(B):
                dets[b], svs[b] = mx_diag(
                    raw, q[b], dof, gen_kwargs.get("ref_frame", "EE"), gen_kwargs.get("xyz_offset")
                )
            out[f"{key}_det"], out[f"{key}_sv"] = dets, svs
        d = np.max(np.abs(res["S"][0] - res["D"][0]), axis=-1) / np.max(np.abs(res["D"][0]), axis=-1)
        print(f"  {key}: B={B} S-vs-D rel median={np.median(d):.2e} p99={np.percentile(d, 99):.2e}", flush=True)

    def run_simple(key, B, seed, make, nt, call):
        rng = np.random.RandomState(seed)
        q, dq, tgt = draw(rng, B, n, nt)
        out[f"{key}_q"], out[f"{key}_dq"], out[f"{key}_target"] = q, dq, tgt
        for label, cfg in (("S", rc), ("D", raw)):
            c = make(cfg)
            out[f"{key}_u{label}"] = np.array([call(c, q[b], dq[b], tgt[b]) for b in range(B)])
        print(f"  {key}: B={B}", flush=True)

    if arm == "twojoint":
        xy = [True, True, False, False, False, False]
        # closed loop = the examples' hot loop (examples/PyGame/force_osc_xy.py:57-78): OSC.generate
        # -> ArmSim._step (arms/twojoint/arm_sim.py:101-137), controller settings of that example
        from abr_control.arms.twojoint import ArmSim

        rng = np.random.RandomState(40)
        Br, T, every = 48, 300, 25
        q0 = np.asarray(rc.START_ANGLES, float) + rng.uniform(-0.6, 0.6, (Br, 2))
        dq0 = rng.uniform(-0.5, 0.5, (Br, 2))
        tgt = np.zeros((Br, 6))
        tgt[:, :2] = rng.uniform(-1.6, 1.6, (Br, 2))
        out["rollout_q0"], out["rollout_dq0"], out["rollout_target"] = q0, dq0, tgt
        out["rollout_every"], out["rollout_T"] = every, T
        for label, cfg in (("S", rc), ("D", raw)):
            qs = np.zeros((Br, T // every, 2))
            dqs = np.zeros((Br, T // every, 2))
            us = np.zeros((Br, T // every, 2))
            for b in range(Br):
                ctrlr = OSC(cfg, kp=20, use_C=True, ctrlr_dof=xy, null_controllers=[
                    Damping(cfg, kv=10),
                    RestingConfig(cfg, kp=50, kv=np.sqrt(50), rest_angles=[np.pi / 4, np.pi])])
                sim = ArmSim(rc, dt=0.001, q_init=q0[b].copy())
                sim.dq = dq0[b].copy()
                for t in range(T):
                    u = ctrlr.generate(q=sim.q, dq=sim.dq, target=tgt[b])
                    sim.send_forces(u)
                    if (t + 1) % every == 0:
                        qs[b, t // every], dqs[b, t // every], us[b, t // every] = sim.q, sim.dq, u
            out[f"rollout_q{label}"], out[f"rollout_dq{label}"], out[f"rollout_u{label}"] = qs, dqs, us
        d = np.max(np.abs(out["rollout_qS"] - out["rollout_qD"]))
        print(f"  rollout: B={Br} T={T}: max |q_S - q_D| over checkpoints = {d:.2e}", flush=True)
        # BASELINE config 1: twojoint OSC position control (row 0 is the "batch=1" case)
        run_osc("cfg1", 256, 0, dict(kp=10, kv=3, ctrlr_dof=xy))
        run_osc("osc_xy_vmax", 128, 2, dict(kp=20, kv=5, ctrlr_dof=xy, vmax=[0.5, 0.5]))
        run_osc("osc_xy_C", 128, 3, dict(kp=10, kv=3, ctrlr_dof=xy, use_C=True))
        run_simple("sliding", 256, 4, lambda c: Sliding(c), 3,
                   lambda c, q, dq, t: c.generate(q, dq, t))
    elif arm == "onejoint":
        # N_LINKS = 1 (arms/onejoint/config.py:32): only link0 - which no joint moves - is summed, so M, g, C are
        # identically zero and OSC (inv(M), osc.py:136) raises in the reference.  What the arm offers is its
        # kinematics (above, every frame) and the controllers that never invert M:
        run_simple("sliding", 128, 60, lambda c: Sliding(c), 3,
                   lambda c, q, dq, t: c.generate(q, dq, t))
        run_simple("sliding_tv", 128, 61, lambda c: Sliding(c, kd=20.0, lamb=5.0), 3,
                   lambda c, q, dq, t: c.generate(q, dq, t, target_velocity=t[::-1] * 0.3, target_acc=t * 0.1))
    elif arm == "threejoint":
        # BASELINE config 5: threejoint Sliding(), defaults kd=160, lamb=30, cartesian
        run_simple("cfg5", 2048, 0, lambda c: Sliding(c), 3,
                   lambda c, q, dq, t: c.generate(q, dq, t))
        run_simple("sliding_tv", 128, 5, lambda c: Sliding(c, kd=20.0, lamb=5.0), 3,
                   lambda c, q, dq, t: c.generate(q, dq, t, target_velocity=t[::-1] * 0.3,
                                                  target_acc=t * 0.1))
        run_osc("osc_xy", 256, 1, dict(kp=50, kv=None, ctrlr_dof=[True, True, False, False, False, False]))
        # planar arm: position xy + orientation about z (examples' 3-dof task)
        for alg in (0, 1):
            run_osc(f"osc_xyg_alg{alg}", 128, 6 + alg,
                    dict(kp=50, ko=20, kv=8, ctrlr_dof=[True, True, False, False, False, True],
                         orientation_algorithm=alg))
        run_simple("joint", 128, 9, lambda c: Joint(c, kp=20, kv=4), 3,
                   lambda c, q, dq, t: c.generate(q, dq, t))
    elif arm == "ur5":
        # BASELINE config 2: UR5 OSC, batch 4096, xyz, use_g
        run_osc("cfg2", 4096, 0, dict(kp=200, ctrlr_dof=xyz))
        # BASELINE config 4 (sample): UR5 OSC + gravity + Coriolis
        run_osc("cfg4", 2048, 1, dict(kp=200, ctrlr_dof=xyz, use_g=True, use_C=True))
        # the reference benchmark's own UR5 setting (examples/timing_plots.py:36)
        for alg in (0, 1):
            run_osc(f"osc6_alg{alg}", 512, 10 + alg,
                    dict(kp=200, ko=150, kv=25, ctrlr_dof=six, orientation_algorithm=alg))
        run_osc("osc_xyz_vmax_ki", 64, 12, dict(kp=100, kv=15, ki=0.2, ctrlr_dof=xyz, vmax=[0.5, 1.0]), steps=5)
        run_osc("osc6_vmax", 128, 13, dict(kp=100, ko=80, kv=15, ctrlr_dof=six, vmax=[0.5, 1.0]))
        run_osc("osc_xyz_tvel", 256, 14, dict(kp=200, ctrlr_dof=xyz), tv=True)
        run_osc("osc_nog", 128, 15, dict(kp=30, kv=7, ctrlr_dof=xyz, use_g=False))
        run_osc("osc_offset", 128, 16, dict(kp=200, ctrlr_dof=xyz),
                gen_kwargs=dict(xyz_offset=np.array([0.11, -0.23, 0.37])))
        run_osc("osc_link5", 128, 17, dict(kp=200, ctrlr_dof=xyz), gen_kwargs=dict(ref_frame="link5"))
        run_osc("osc_abg", 128, 18, dict(kp=100, ko=60, kv=12,
                                          ctrlr_dof=[False, False, False, True, True, True]))
        run_osc("osc_xz_b", 128, 19, dict(kp=100, ko=60, kv=12,
                                           ctrlr_dof=[True, False, True, False, True, False]))
        run_osc("osc_null2", 256, 20, dict(kp=200, ctrlr_dof=xyz),
                null=[(Damping, dict(kv=10)),
                      (RestingConfig, dict(rest_angles=[None, 0.8, -1.6, None, 1.5, None], kp=40, kv=8))])
        run_simple("joint", 256, 21, lambda c: Joint(c, kp=50, kv=9), 6,
                   lambda c, q, dq, t: c.generate(q, dq, t * 3.0))
        run_simple("joint_tv_nog", 128, 22, lambda c: Joint(c, kp=50, account_for_gravity=False), 6,
                   lambda c, q, dq, t: c.generate(q, dq, t * 3.0, target_velocity=t[::-1]))
        run_simple("sliding", 256, 23, lambda c: Sliding(c), 3,
                   lambda c, q, dq, t: c.generate(q, dq, t))
    elif arm == "jaco2":
        # BASELINE config 3 (sample of the 16384): Jaco2 OSC + null-space Damping
        run_osc("cfg3", 2048, 0, dict(kp=200, ctrlr_dof=xyz), null=[(Damping, dict(kv=10))])
        # the reference benchmark's Jaco2 setting (examples/timing_plots.py:37)
        run_osc("osc5", 512, 30, dict(kp=200, ctrlr_dof=[True] * 5 + [False]))
        run_osc("osc6_alg1", 256, 31, dict(kp=200, ko=150, kv=25, ctrlr_dof=six, orientation_algorithm=1))
        run_osc("osc_rest", 256, 32, dict(kp=200, ctrlr_dof=xyz),
                null=[(RestingConfig, dict(rest_angles=[None, 3.14, 1.57, None, None, 3.04], kp=30, kv=6))])
        run_simple("damping", 128, 33, lambda c: Damping(c, kv=10), 6,
                   lambda c, q, dq, t: c.generate(q, dq))

    if arm in ("ur5", "jaco2"):
        # iterative inverse kinematics (controllers/path_planners/inverse_kinematics.py:28-135)
        from abr_control.controllers.path_planners.inverse_kinematics import InverseKinematics
        from abr_control.utils import transformations as tf

        rng = np.random.RandomState(50)
        Bi, T = 6, 200
        q0 = rng.uniform(0.3, 2.8, (Bi, n))
        qg = q0 + rng.uniform(-0.8, 0.8, (Bi, n))  # reachable goals: pose of a nearby configuration
        tgt = np.zeros((Bi, 6))
        for b in range(Bi):
            tgt[b, :3] = rc.Tx("EE", qg[b])
            tgt[b, 3:] = tf.euler_from_quaternion(raw.quaternion("EE", qg[b]), axes="sxyz")
        out["ik_q0"], out["ik_target"] = q0, tgt
        for method in (1, 2, 3):
            for label, cfg in (("S", rc), ("D", raw)):
                pp = np.zeros((Bi, T, n))
                vp = np.zeros((Bi, T, n))
                for b in range(Bi):
                    ik = InverseKinematics(cfg)
                    pp[b], vp[b] = ik.generate_path(q0[b], tgt[b], n_timesteps=T, dt=0.001, method=method)
                out[f"ik_m{method}_pos{label}"], out[f"ik_m{method}_vel{label}"] = pp, vp
            d = np.max(np.abs(out[f"ik_m{method}_posS"] - out[f"ik_m{method}_posD"]))
            print(f"  ik method {method}: max |q_S - q_D| over {T} steps = {d:.2e}", flush=True)

    np.savez_compressed(f"{OUT}/{arm}.npz", **out)
    print(f"  wrote {OUT}/{arm}.npz ({len(out)} arrays)", flush=True)


# ----------------------------------------------------------------------------------------
def gen_known():
    """Closed-form twojoint answers the reference's own tests pin its functions against
    (abr_control/arms/tests/dummy_base_arm.py; grids as in test_base_config.py:40-180,
    coarser for the 4-D (q,dq) grids to keep the fixture small), plus the doctest
    constants of utils/transformations.py and the exact values of test_osc.py:12-59."""
    from abr_control.arms.tests.dummy_base_arm import TwoJoint
    from abr_control.utils import transformations as tf

    ta = TwoJoint()
    out = {}
    qv = np.linspace(0, 2 * np.pi, 25)
    Q = np.array([[a, b] for a in qv for b in qv])
    out["q_grid"] = Q
    for f in ("link0", "joint0", "link1", "joint1", "link2", "EE"):
        out[f"R_{f}"] = np.array([getattr(ta, f"R_{f}")(q) for q in Q], dtype=float)
        out[f"Tx_{f}"] = np.array([getattr(ta, f"Tx_{f}")(q) for q in Q], dtype=float)
        out[f"Tinv_{f}"] = np.array([getattr(ta, f"T_inv_{f}")(q) for q in Q], dtype=float)
        out[f"J_{f}"] = np.array([getattr(ta, f"J_{f}")(q) for q in Q], dtype=float)
    out["M"] = np.array([ta.M(q) for q in Q], dtype=float)
    out["g"] = np.array([ta.g(q) for q in Q], dtype=float)
    v = np.linspace(0, 2 * np.pi, 7)
    QD = np.array([[a, b, c, d] for a in v for b in v for c in v for d in v])
    out["qdq_grid"] = QD
    for f in ("link0", "joint0", "link1", "joint1", "link2", "EE"):
        out[f"dJ_{f}"] = np.array([getattr(ta, f"dJ_{f}")(x[:2], x[2:]) for x in QD], dtype=float)
    out["C"] = np.array([ta.C(x[:2], x[2:]) for x in QD], dtype=float)

    # transformations.py known answers (run the reference's functions on seeded inputs)
    rng = np.random.RandomState(0)
    ang = rng.uniform(-2 * np.pi, 2 * np.pi, (64, 3))
    out["tf_angles"] = ang
    out["tf_quat_from_euler_rxyz"] = np.array([tf.quaternion_from_euler(*a, axes="rxyz") for a in ang])
    out["tf_euler_matrix_rxyz"] = np.array([tf.euler_matrix(*a, axes="rxyz")[:3, :3] for a in ang])
    out["tf_quat_from_matrix"] = np.array(
        [tf.quaternion_from_matrix(tf.euler_matrix(*a, axes="rxyz")) for a in ang])
    qa = rng.normal(size=(64, 4))
    qb = rng.normal(size=(64, 4))
    out["tf_qa"], out["tf_qb"] = qa, qb
    out["tf_quat_mul"] = np.array([tf.quaternion_multiply(a, b) for a, b in zip(qa, qb)])
    out["tf_quat_conj"] = np.array([tf.quaternion_conjugate(a) for a in qa])
    out["tf_unit"] = np.array([tf.unit_vector(a) for a in qa])
    np.savez_compressed(f"{OUT}/known_answers.npz", **out)
    print(f"  wrote {OUT}/known_answers.npz", flush=True)


# ----------------------------------------------------------------------------------------
def limit_sets(n):
    """two AvoidJointLimits parameterisations covering walls / gradients / zero-crossing ranges / no limits"""
    nan = np.nan
    a = dict(mn=[], mx=[], cz=[], gr=[], mt=[])
    b = dict(mn=[], mx=[], cz=[], gr=[], mt=[])
    for i in range(n):
        k = i % 4
        a["mn"].append([np.pi / 5.0, 1.0, 5.5, nan][k])
        a["mx"].append([np.pi / 2.0, 4.0, 0.8, nan][k])
        a["cz"].append([False, False, True, False][k])
        a["gr"].append([False, True, False, False][k])
        a["mt"].append([100.0, 7.5, 3.0, 1.0][k])
        b["mn"].append([5.0, nan, 0.4, 2.0][k])
        b["mx"].append([1.2, 3.0, nan, 2.5][k])
        b["cz"].append([True, False, False, False][k])
        b["gr"].append([True, True, False, True][k])
        b["mt"].append([2.0, 0.5, 4.0, 1e9][k])
    return {"limA": a, "limB": b}


OBSTACLES = {
    # [x, y, z, radius] inside each arm's workspace; threshold / gain as in examples/PyGame/avoid_obstacles.py:17
    "twojoint": dict(obstacles=[[1.0, 1.0, 0, 0.2], [-0.5, 1.5, 0, 0.2], [0.5, -1.0, 0, 0.3]], threshold=1, gain=30),
    "threejoint": dict(obstacles=[[1.0, 1.0, 0, 0.2], [-0.5, 1.5, 0, 0.2], [0.5, -1.0, 0, 0.3]], threshold=1, gain=30),
    "ur5": dict(obstacles=[[0.3, 0.2, 0.4, 0.1], [-0.2, 0.4, 0.3, 0.05], [0.1, -0.3, 0.6, 0.15]], threshold=0.3,
                gain=30),
    "jaco2": dict(obstacles=[[0.3, 0.2, 0.4, 0.1], [-0.2, 0.4, 0.3, 0.05], [0.1, -0.3, 0.6, 0.15]], threshold=0.3,
                  gain=30),
}


def gen_secondary(arm):
    """AvoidJointLimits / Floating / AvoidObstacles of the reference on seeded states."""
    mod = importlib.import_module(f"abr_control.arms.{arm}")
    from abr_control.controllers import AvoidJointLimits, AvoidObstacles, Floating

    rc = mod.Config(use_cython=True)
    raw = RawConfig(rc)
    n = rc.N_JOINTS
    out = {}
    rng = np.random.RandomState(70)
    B = 256
    q, dq, _ = draw(rng, B, n)
    out["lim_q"] = q
    for key, ps in limit_sets(n).items():
        c = AvoidJointLimits(rc, min_joint_angles=list(ps["mn"]), max_joint_angles=list(ps["mx"]),
                             max_torque=list(ps["mt"]), cross_zero=list(ps["cz"]), gradient=list(ps["gr"]))
        with np.errstate(all="ignore"):
            out[f"{key}_u"] = np.array([c.generate(q[b], dq[b]) for b in range(B)])
        for f in ("mn", "mx", "mt", "cz", "gr"):
            out[f"{key}_{f}"] = np.array(ps[f])
        print(f"  {key}: nonzero rows {np.count_nonzero(np.any(out[f'{key}_u'] != 0, axis=1))}/{B}", flush=True)

    Bf = 128
    q, dq, _ = draw(np.random.RandomState(71), Bf, n)
    out["float_q"], out["float_dq"] = q, dq
    for dyn in (0, 1):
        for ts in (0, 1):
            for label, cfg in (("S", rc), ("D", raw)):
                c = Floating(cfg, dynamic=bool(dyn), task_space=bool(ts))
                out[f"float_d{dyn}t{ts}_u{label}"] = np.array(
                    [np.asarray(c.generate(q[b], dq[b]), dtype=float) for b in range(Bf)])
            d = np.max(np.abs(out[f"float_d{dyn}t{ts}_uS"] - out[f"float_d{dyn}t{ts}_uD"]), axis=1) / np.maximum(
                np.max(np.abs(out[f"float_d{dyn}t{ts}_uD"]), axis=1), 1e-300)
            print(f"  floating dynamic={dyn} task_space={ts}: S-vs-D rel median={np.median(d):.2e} "
                  f"max={np.max(d):.2e}", flush=True)

    Bo = 128
    q, _, _ = draw(np.random.RandomState(72), Bo, n)
    out["obs_q"] = q
    kw = OBSTACLES[arm]
    out["obs_obstacles"] = np.array(kw["obstacles"], dtype=float)
    out["obs_threshold"], out["obs_gain"] = float(kw["threshold"]), float(kw["gain"])
    for label, cfg in (("S", rc), ("D", raw)):
        c = AvoidObstacles(cfg, **kw)
        with np.errstate(all="ignore"):
            out[f"obs_u{label}"] = np.array([c.generate(q[b]) for b in range(Bo)])
            if label == "D":  # the same signal before np.clip (avoid_obstacles.py:121): the scale errors live on
                c.maximum = 1e300
                out["obs_uD_unclipped"] = np.array([c.generate(q[b]) for b in range(Bo)])
    act = np.any(out["obs_uD"] != 0, axis=1)
    d = np.max(np.abs(out["obs_uS"] - out["obs_uD"]), axis=1)[act] / np.max(np.abs(out["obs_uD"]), axis=1)[act]
    print(f"  obstacles: active rows {act.sum()}/{Bo}; S-vs-D rel median={np.median(d):.2e} max={np.max(d):.2e}",
          flush=True)
    # the three of them (+ Damping) behind OSC's null-space filter, as in
    # examples/PyGame/force_osc_xy_avoid_joint_limits.py:21-36 and avoid_obstacles.py:17-28
    from abr_control.controllers import OSC, Damping

    dof = [True, True, n > 3, False, False, False]
    Bs = 128
    q, dq, tgt = draw(np.random.RandomState(73), Bs, n)
    out["oscsec_q"], out["oscsec_dq"], out["oscsec_target"] = q, dq, tgt
    out["oscsec_dof"] = np.array(dof)
    ps = limit_sets(n)["limA"]
    for label, cfg in (("S", rc), ("D", raw)):
        nulls = [
            AvoidJointLimits(cfg, min_joint_angles=list(ps["mn"]), max_joint_angles=list(ps["mx"]),
                             max_torque=list(ps["mt"]), cross_zero=list(ps["cz"]), gradient=list(ps["gr"])),
            AvoidObstacles(cfg, **kw),
            Damping(cfg, kv=10),
        ]
        c = OSC(cfg, kp=100, null_controllers=nulls, ctrlr_dof=dof)
        with np.errstate(all="ignore"):
            out[f"oscsec_u{label}"] = np.array([c.generate(q[b], dq[b], tgt[b]) for b in range(Bs)])
    dets, svs = np.zeros(Bs), np.zeros((Bs, int(np.sum(dof))))
    for b in range(Bs):
        dets[b], svs[b] = mx_diag(raw, q[b], dof)
    out["oscsec_det"], out["oscsec_sv"] = dets, svs
    d = np.max(np.abs(out["oscsec_uS"] - out["oscsec_uD"]), axis=1) / np.max(np.abs(out["oscsec_uD"]), axis=1)
    print(f"  OSC + [limits, obstacles, damping]: S-vs-D rel median={np.median(d):.2e} max={np.max(d):.2e}", flush=True)
    np.savez_compressed(f"{OUT}/sec_{arm}.npz", **out)
    print(f"  wrote {OUT}/sec_{arm}.npz ({len(out)} arrays)", flush=True)


def gen_osc_helpers(arm):
    """The three helper methods of OSC that the reference's own tests call directly
    (controllers/tests/test_osc.py:12-59 `_velocity_limiting`, :62-86 `_Mx`, :94-140
    `_calc_orientation_forces`), run on the reference classes: the inputs of those tests plus seeded random ones."""
    mod = importlib.import_module(f"abr_control.arms.{arm}")
    from abr_control.controllers import OSC

    rc = mod.Config(use_cython=True)
    raw = RawConfig(rc)
    n = rc.N_JOINTS
    out = {}
    # ---- _velocity_limiting: test_osc.py:24-59 (kp=10, ko=8, kv=4, vmax=[1,1]) + random task-space errors
    kp, ko, kv, vmax = 10, 8, 4, 1
    c = OSC(raw, kp=kp, ko=ko, kv=kv, vmax=[vmax, vmax], ctrlr_dof=[True] * 6)
    rng = np.random.RandomState(80)
    ut = np.vstack([np.ones((1, 6)) * 0.05, np.hstack([np.ones((1, 3)) * 100, np.ones((1, 3)) * 0.05]),
                    np.ones((1, 6)) * 100, rng.uniform(-1, 1, (61, 6)) * rng.choice([0.05, 0.5, 5.0], (61, 1))])
    out["vl_gains"] = np.array([kp, ko, kv, vmax, vmax], dtype=float)
    out["vl_in"] = ut
    out["vl_out"] = np.array([c._velocity_limiting(u.copy()) for u in ut])
    # the expected values the reference test asserts (test_osc.py:36-59)
    assert np.allclose(out["vl_out"][0], [kp * 0.05] * 3 + [ko * 0.05] * 3, atol=1e-5)
    assert np.allclose(out["vl_out"][1], [kv * np.sqrt(vmax / 3.0)] * 3 + [ko * 0.05] * 3, atol=1e-5)
    assert np.allclose(out["vl_out"][2], [kv * np.sqrt(vmax / 3.0)] * 6, atol=1e-5)
    # ---- _Mx: test_osc.py:62-86 (J = I with threshold 1e-5; J = ones) + random selected rows
    c = OSC(raw, ctrlr_dof=[True] * 6 if n >= 6 else [True] * n + [False] * (6 - n))
    B = 100
    q = rng.uniform(0, 2 * np.pi, (B, n))
    M = np.array([raw.M(q[b]) for b in range(B)])
    out["mx_q"], out["mx_M"] = q, M
    r = [c._Mx(M=M[b], J=np.eye(n), threshold=1e-5) for b in range(B)]
    out["mx_eye_Mx"], out["mx_eye_Minv"] = np.array([x[0] for x in r]), np.array([x[1] for x in r])
    assert all(np.allclose(M[b], out["mx_eye_Mx"][b], atol=1e-5) for b in range(B))  # test_osc.py:80
    r = [c._Mx(M=M[b], J=np.ones((6, n))) for b in range(B)]
    out["mx_ones_Mx"] = np.array([x[0] for x in r])
    assert all(np.all(np.abs(np.linalg.svd(x)[1][1:]) < 1e-10) for x in out["mx_ones_Mx"])  # test_osc.py:86
    for k in sorted({1, 2, 3, min(n, 6)}):
        if k > 6:
            continue
        Jk = np.array([raw.J("EE", q[b])[rng.permutation(6)[:k]] for b in range(B)])
        r = [c._Mx(M=M[b], J=Jk[b]) for b in range(B)]
        out[f"mx_k{k}_J"] = Jk
        out[f"mx_k{k}_Mx"] = np.array([x[0] for x in r])
        A = np.array([Jk[b] @ np.linalg.inv(M[b]) @ Jk[b].T for b in range(B)])
        out[f"mx_k{k}_det"] = np.linalg.det(A)
        out[f"mx_k{k}_sv"] = np.linalg.svd(A, compute_uv=False)
    # ---- _calc_orientation_forces (osc.py:149-196; the reference test calls it without ref_frame,
    # test_osc.py:127, and fails for that reason - called here with the argument the method requires)
    Bo = 100
    q = rng.uniform(0, 2 * np.pi, (Bo, n))
    abg = rng.uniform(-np.pi, np.pi, (Bo, 3))
    out["of_q"], out["of_abg"] = q, abg
    out["of_R"] = np.array([raw.R("EE", q[b]) for b in range(Bo)])
    for alg in (0, 1):
        c = OSC(raw, orientation_algorithm=alg, ctrlr_dof=[True] * 6)
        out[f"of_alg{alg}"] = np.array([c._calc_orientation_forces(abg[b], q[b], "EE") for b in range(Bo)])
    np.savez_compressed(f"{OUT}/oschelpers_{arm}.npz", **out)
    print(f"  wrote {OUT}/oschelpers_{arm}.npz ({len(out)} arrays)", flush=True)


def gen_quat(arm):
    """robot_config.R / .quaternion (base_config.py:287-318; quaternion_from_matrix = eigh of the 4x4 K matrix,
    transformations.py:1192-1271) for EVERY frame: Jaco2's rounded rotation constants make R R^T - I grow to 4e-4
    along the chain, which is where a non-eigh quaternion extraction could drift from the reference's."""
    mod = importlib.import_module(f"abr_control.arms.{arm}")
    rc = mod.Config(use_cython=True)
    raw = RawConfig(rc)
    n = rc.N_JOINTS
    rng = np.random.RandomState(90)
    B = 64
    q = rng.uniform(0, 2 * np.pi, (B, n))
    frames = frames_of(rc)
    out = {"q": q, "frames": np.array(frames)}
    for f in frames:
        out[f"R_{f}"] = np.array([raw.R(f, q[b]) for b in range(B)])
        out[f"quat_{f}"] = np.array([raw.quaternion(f, q[b]) for b in range(B)])
        dev = max(np.max(np.abs(R @ R.T - np.eye(3))) for R in out[f"R_{f}"])
        print(f"  {f}: max |R R^T - I| = {dev:.1e}", flush=True)
    np.savez_compressed(f"{OUT}/quat_{arm}.npz", **out)
    print(f"  wrote {OUT}/quat_{arm}.npz ({len(out)} arrays)", flush=True)


if what == "known":
    gen_known()
elif what.startswith("quat:"):
    gen_quat(what[5:])
elif what.startswith("helpers:"):
    gen_osc_helpers(what[8:])
elif what.startswith("sec:"):
    gen_secondary(what[4:])
else:
    gen_arm(what)
