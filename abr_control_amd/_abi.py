"""ctypes mirror of include/abrk.h (plain data only) + arm-table helpers.

Kept free of any device code so that tools and tests can build `abrk_arm_desc` /
`abrk_osc_params` values without touching the GPU library.
"""
import ctypes as C
import json
import os

import numpy as np

MAX_JOINTS = 7
MAX_NULL = 4

F64, F32 = 0, 1
NULL_DAMPING, NULL_RESTING = 1, 2

WANT_TX = 1 << 0
WANT_J = 1 << 1
WANT_M = 1 << 2
WANT_G = 1 << 3
WANT_C = 1 << 4
WANT_DJ = 1 << 5
WANT_R = 1 << 6
WANT_T = 1 << 7
WANT_TINV = 1 << 8
WANT_QUAT = 1 << 9

ERRORS = {-1: "EINVAL", -2: "ENODEV", -3: "ENOMEM", -4: "ENOARM", -5: "EFRAME", -6: "ESINGULAR"}
ESINGULAR = -6


class ArmDesc(C.Structure):
    _fields_ = [
        ("n_joints", C.c_int32),
        ("n_links_dyn", C.c_int32),
        ("has_ee", C.c_int32),
        ("reserved", C.c_int32),
        ("A0", C.c_double * 12),
        ("AJ", (C.c_double * 12) * MAX_JOINTS),
        ("B", (C.c_double * 12) * MAX_JOINTS),
        ("E", C.c_double * 12),
        ("mdiag", (C.c_double * 6) * (MAX_JOINTS + 1)),
        ("name", C.c_char * 32),
    ]


class DynOut(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("Tx", "J", "M", "g", "C", "dJ", "R", "T", "Tinv", "quat")]


class ScratchInfo(C.Structure):  # abrk_scratch_info
    _fields_ = [(k, C.c_int64) for k in ("worklist_slots", "worklist_bytes", "inline_fallbacks", "evictions",
                                          "device_free_bytes", "device_total_bytes", "status_words_out",
                                          "status_blocks")]


class ShardCut(C.Structure):  # abrk_shard_cut: how a resident batch is cut over the devices
    _fields_ = [("n_shards", C.c_int32), ("devices", C.POINTER(C.c_int32)), ("rows", C.POINTER(C.c_int64)),
                ("streams", C.POINTER(C.c_void_p))]


class NullCtrl(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("rest_mask", C.c_int32 * MAX_JOINTS),
        ("kp", C.c_double),
        ("kv", C.c_double),
        ("rest_angles", C.c_double * MAX_JOINTS),
    ]


class OSCParams(C.Structure):
    _fields_ = [
        ("kp", C.c_double),
        ("ko", C.c_double),
        ("kv", C.c_double),
        ("ki", C.c_double),
        ("use_vmax", C.c_int32),
        ("use_g", C.c_int32),
        ("use_C", C.c_int32),
        ("orientation_algorithm", C.c_int32),
        ("vmax", C.c_double * 2),
        ("ctrlr_dof", C.c_int32 * 6),
        ("ref_frame", C.c_int32),
        ("n_null", C.c_int32),
        ("xyz_offset", C.c_double * 3),
        ("null_ctrl", NullCtrl * MAX_NULL),
    ]


class SlidingParams(C.Structure):
    _fields_ = [
        ("kd", C.c_double),
        ("lamb", C.c_double),
        ("cartesian", C.c_int32),
        ("ref_frame", C.c_int32),
        ("offset", C.c_double * 3),
    ]


class IkParams(C.Structure):
    _fields_ = [("max_dx", C.c_double), ("max_dr", C.c_double), ("max_dq", C.c_double), ("dt", C.c_double),
                ("n_timesteps", C.c_int32), ("method", C.c_int32)]


def make_ik_params(max_dx=0.2, max_dr=2 * np.pi, max_dq=np.pi, n_timesteps=200, dt=0.001, method=3):
    """InverseKinematics(max_dx, max_dr, max_dq).generate_path(n_timesteps, dt, method) defaults
    (controllers/path_planners/inverse_kinematics.py:21-37)."""
    p = IkParams()
    p.max_dx, p.max_dr, p.max_dq, p.dt = max_dx, max_dr, max_dq, dt
    p.n_timesteps, p.method = int(n_timesteps), int(method)
    return p


MAX_OBSTACLES = 16


class LimitsParams(C.Structure):
    _fields_ = [
        ("min_joint_angles", C.c_double * MAX_JOINTS),
        ("max_joint_angles", C.c_double * MAX_JOINTS),
        ("max_torque", C.c_double * MAX_JOINTS),
        ("cross_zero", C.c_int32 * MAX_JOINTS),
        ("gradient", C.c_int32 * MAX_JOINTS),
        ("no_limits_min", C.c_int32 * MAX_JOINTS),
        ("no_limits_max", C.c_int32 * MAX_JOINTS),
    ]


class ObstaclesParams(C.Structure):
    _fields_ = [
        ("n_obstacles", C.c_int32),
        ("reserved", C.c_int32),
        ("threshold", C.c_double),
        ("gain", C.c_double),
        ("maximum", C.c_double),
        ("obstacles", (C.c_double * 4) * MAX_OBSTACLES),
    ]


def make_limits_params(n_joints, min_joint_angles, max_joint_angles, max_torque=None, cross_zero=None,
                       gradient=None):
    """AvoidJointLimits.__init__ (avoid_joint_limits.py:35-81) -> abrk_limits_params: limits shifted by
    -pi (:45-50), swapped where cross_zero (:62-66), 'no limit' = NaN (or None) flagged (:74-75)."""
    n = int(n_joints)
    mn = np.array([np.nan if v is None else float(v) for v in min_joint_angles], dtype=float) - np.pi
    mx = np.array([np.nan if v is None else float(v) for v in max_joint_angles], dtype=float) - np.pi
    if mn.shape[0] != n or mx.shape[0] != n:
        raise Exception("joint angles vector incorrect size")  # avoid_joint_limits.py:68-72
    cz = np.zeros(n, bool) if cross_zero is None else np.array(cross_zero, dtype=bool)
    gr = np.zeros(n, bool) if gradient is None else np.array(gradient, dtype=bool)
    tmin, tmax = mn.copy(), mx.copy()
    mx[cz], mn[cz] = tmin[cz], tmax[cz]
    mt = np.ones(n) if max_torque is None else np.asarray(max_torque, dtype=float)
    p = LimitsParams()
    for i in range(n):
        p.no_limits_min[i], p.no_limits_max[i] = int(np.isnan(mn[i])), int(np.isnan(mx[i]))
        p.min_joint_angles[i] = 0.0 if np.isnan(mn[i]) else mn[i]
        p.max_joint_angles[i] = 0.0 if np.isnan(mx[i]) else mx[i]
        p.max_torque[i] = mt[i]
        p.cross_zero[i], p.gradient[i] = int(cz[i]), int(gr[i])
    return p


def make_obstacles_params(obstacles=None, threshold=0.2, gain=1, maximum=500):
    """AvoidObstacles.__init__ / set_obstacles (avoid_obstacles.py:26-36,122-133) -> abrk_obstacles_params."""
    obs = np.zeros((0, 4)) if obstacles is None or len(obstacles) == 0 else np.asarray(obstacles, dtype=float)
    if obs.ndim != 2 or obs.shape[1] != 4:
        raise ValueError("obstacles must be a list of [x, y, z, radius]")
    if obs.shape[0] > MAX_OBSTACLES:
        raise ValueError(f"at most {MAX_OBSTACLES} obstacles")
    p = ObstaclesParams()
    p.n_obstacles = obs.shape[0]
    p.threshold, p.gain, p.maximum = float(threshold), float(gain), float(maximum)
    for i in range(obs.shape[0]):
        for r in range(4):
            p.obstacles[i][r] = obs[i, r]
    return p


class TwoLinkPlant(C.Structure):
    _fields_ = [("K1", C.c_double), ("K2", C.c_double), ("K3", C.c_double), ("K4", C.c_double), ("dt", C.c_double)]


def make_twolink_plant(L, M_LINKS, dt=0.001):
    """The constants ArmSim.__init__ derives from robot_config.L / _M_LINKS
    (abr_control/arms/twojoint/arm_sim.py:26-41), expression for expression."""
    L = np.asarray(L, dtype=float)
    M = M_LINKS
    Ls = [np.sum(L[ii * 2: ii * 2 + 2]) for ii in range(int(L.shape[0] / 2))]
    p = TwoLinkPlant()
    p.K1 = (1 / 3.0 * M[1][0, 0] + M[2][0, 0]) * Ls[1] ** 2.0 + 1 / 3.0 * M[2][0, 0] * Ls[2] ** 2.0
    p.K2 = M[2][0, 0] * Ls[1] * Ls[2]
    p.K3 = 1 / 3.0 * M[2][0, 0] * Ls[2] ** 2.0
    p.K4 = 1 / 2.0 * M[2][0, 0] * Ls[1] * Ls[2]
    p.dt = dt
    return p


TABLE_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "arms", "tables")
BUILTIN_ARMS = ("ur5", "jaco2", "twojoint", "threejoint", "onejoint")


def load_table(name):
    """Arm table (dict) of a built-in arm; see tools/extract_arm_table.py for provenance."""
    with open(os.path.join(TABLE_DIR, f"{name}.json")) as fh:
        return json.load(fh)


def desc_from_table(tab):
    n = int(tab["n_joints"])
    if not 1 <= n <= MAX_JOINTS:
        raise ValueError(f"n_joints={n} outside 1..{MAX_JOINTS}")
    d = ArmDesc()
    d.n_joints = n
    d.n_links_dyn = int(tab["n_links_dyn"])
    d.has_ee = int(tab["has_ee"])
    ident = [1.0, 0, 0, 0, 0, 1.0, 0, 0, 0, 0, 1.0, 0]

    def put(dst, m):
        flat = np.asarray(m, dtype=np.float64).reshape(12)
        for i in range(12):
            dst[i] = flat[i]

    put(d.A0, tab["A0"])
    for i in range(MAX_JOINTS):
        put(d.AJ[i], tab["AJ"][i] if i < n else ident)
        put(d.B[i], tab["B"][i] if i < n else ident)
    put(d.E, tab["E"])
    for l in range(MAX_JOINTS + 1):
        row = tab["mdiag"][l] if l < len(tab["mdiag"]) else [0.0] * 6
        for r in range(6):
            d.mdiag[l][r] = float(row[r])
    d.name = tab.get("name", "robot").encode()[:31]
    return d


def render_tab_struct(t, struct_name, name=None):
    """C++ source of the constexpr arm table `struct <struct_name>` the StaticArm kernels are instantiated on
    (abrk_arms_builtin.h for the built-in arms, the plugin source of a compiled user arm).  Literals are `repr`
    of the doubles, i.e. they read back to exactly the values of the table."""
    def lit(v):
        return repr(float(v))

    def mat(m):
        return "{" + ", ".join(lit(v) for row in m for v in row) + "}"

    ident = [[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0]]
    n = int(t["n_joints"])
    md = [list(r) for r in t["mdiag"]] + [[0.0] * 6] * (n + 1 - len(t["mdiag"]))
    out = [
        f"struct {struct_name} {{",
        f"  static constexpr int N = {n};",
        f"  static constexpr int NL = {int(t['n_links_dyn'])};",
        f"  static constexpr bool kHasEE = {'true' if t['has_ee'] else 'false'};",
        f"  static constexpr double A0[12] = {mat(t['A0'])};",
        f"  static constexpr double AJ[{n}][12] = {{" + ",\n      ".join(mat(m) for m in t["AJ"][:n]) + "};",
        f"  static constexpr double B[{n}][12] = {{" + ",\n      ".join(mat(m) for m in t["B"][:n]) + "};",
        f"  static constexpr double E[12] = {mat(t['E'] if t['has_ee'] else ident)};",
        f"  static constexpr double MD[{n + 1}][6] = {{" + ",\n      ".join(
            "{" + ", ".join(lit(v) for v in row) + "}" for row in md[: n + 1]) + "};",
        f'  static constexpr const char* kName = "{name if name is not None else t.get("name", "robot")}";',
        "};",
    ]
    return "\n".join(out)


def table_from_desc(d):
    n = d.n_joints
    m = lambda a: np.array(list(a), dtype=np.float64).reshape(3, 4).tolist()
    return {
        "name": d.name.decode(),
        "n_joints": n,
        "n_links_dyn": d.n_links_dyn,
        "has_ee": d.has_ee,
        "A0": m(d.A0),
        "AJ": [m(d.AJ[i]) for i in range(n)],
        "B": [m(d.B[i]) for i in range(n)],
        "E": m(d.E),
        "mdiag": [list(d.mdiag[l]) for l in range(n + 1)],
    }


def frame_id(name, n_joints):
    """'link{i}' -> 2i, 'joint{i}' -> 2i+1, 'EE' -> 2n+1; anything else raises exactly
    like the reference (`Exception("Invalid transformation name: ...")`,
    abr_control/arms/ur5/config.py:337)."""
    try:
        if name == "EE":
            return 2 * n_joints + 1
        if name.startswith("link"):
            i = int(name[4:])
            if 0 <= i <= n_joints:
                return 2 * i
        elif name.startswith("joint"):
            i = int(name[5:])
            if 0 <= i < n_joints:
                return 2 * i + 1
    except (ValueError, AttributeError):
        pass
    raise Exception(f"Invalid transformation name: {name}")


def make_osc_params(
    n_joints,
    kp=1,
    ko=None,
    kv=None,
    ki=0,
    vmax=None,
    ctrlr_dof=None,
    null_controllers=(),
    use_g=True,
    use_C=False,
    orientation_algorithm=0,
    ref_frame="EE",
    xyz_offset=None,
):
    """OSC.__init__ defaults (abr_control/controllers/osc.py:53-118) -> abrk_osc_params."""
    p = OSCParams()
    p.kp = kp
    p.ko = kp if ko is None else ko
    p.kv = float(np.sqrt(p.kp + p.ko)) if kv is None else kv
    p.ki = ki
    p.use_vmax = int(vmax is not None)
    if vmax is not None:
        p.vmax[0], p.vmax[1] = float(vmax[0]), float(vmax[1])
    if ctrlr_dof is None:
        ctrlr_dof = [True, True, True, False, False, False]
    for r in range(6):
        p.ctrlr_dof[r] = int(bool(ctrlr_dof[r]))
    p.use_g = int(bool(use_g))
    p.use_C = int(bool(use_C))
    p.orientation_algorithm = int(orientation_algorithm)
    p.ref_frame = frame_id(ref_frame, n_joints)
    if xyz_offset is not None:
        for r in range(3):
            p.xyz_offset[r] = float(xyz_offset[r])
    null_controllers = list(null_controllers or ())
    if len(null_controllers) > MAX_NULL:
        raise ValueError(f"at most {MAX_NULL} fused null controllers")
    p.n_null = len(null_controllers)
    for i, nc in enumerate(null_controllers):
        p.null_ctrl[i] = nc
    return p


def make_damping(kv):
    c = NullCtrl()
    c.kind = NULL_DAMPING
    c.kv = kv
    return c


def make_resting(rest_angles, kp=1, kv=None):
    """RestingConfig(rest_angles, kp, kv) (resting_config.py:18-23; Joint defaults
    joint.py:26-33: kv = sqrt(kp))."""
    c = NullCtrl()
    c.kind = NULL_RESTING
    c.kp = kp
    c.kv = float(np.sqrt(kp)) if kv is None else kv
    for i, v in enumerate(rest_angles):
        c.rest_mask[i] = int(v is not None)
        c.rest_angles[i] = 0.0 if v is None else float(v)
    return c


def make_joint(kp=1, kv=None):
    c = NullCtrl()
    c.kind = 0
    c.kp = kp
    c.kv = float(np.sqrt(kp)) if kv is None else kv
    return c


def make_sliding_params(n_joints, kd=160.0, lamb=30.0, cartesian=True, ref_frame="EE", offset=None):
    p = SlidingParams()
    p.kd, p.lamb = kd, lamb
    p.cartesian = int(bool(cartesian))
    p.ref_frame = frame_id(ref_frame, n_joints)
    if offset is not None:
        for r in range(3):
            p.offset[r] = float(offset[r])
    return p
