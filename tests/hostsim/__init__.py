"""Python side of the hostsim TEST AID (tests/hostsim/hostsim.cpp): same call surface as
abr_control_amd.engine, but runs the row programs on the CPU.  Never imported by the product."""
import ctypes as C
import os
import subprocess

import numpy as np

from abr_control_amd import _abi
from abr_control_amd.engine import _OUT_SHAPES, _WANT_BITS, _dtype_code

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "..", "..", "abr_control_amd", "csrc")
# built in parts, in parallel (a single translation unit takes minutes)
PARTS = {
    "static": ["-DHOSTSIM_STATIC=1"],
    "rt13": ["-DHOSTSIM_RT_LO=1", "-DHOSTSIM_RT_HI=3"],
    "rt45": ["-DHOSTSIM_RT_LO=4", "-DHOSTSIM_RT_HI=5"],
    "rt6": ["-DHOSTSIM_RT_LO=6", "-DHOSTSIM_RT_HI=6"],
    "rt7": ["-DHOSTSIM_RT_LO=7", "-DHOSTSIM_RT_HI=7"],
    "law": ["-DHOSTSIM_LAW=1"],
}
_libs = {}


def _so(part):
    return os.path.join(_HERE, f"libabrk_hostsim_{part}.so")


def build(force=False):
    srcs = [os.path.join(_HERE, "hostsim.cpp")] + [
        os.path.join(_CSRC, f)
        for f in ("abrk_device.h", "abrk_ctrl.h", "abrk_rows.h", "abrk_params.h", "abrk_rt.h", "abrk_arms_builtin.h",
                  "abrk_sincos_table.h")]
    newest = max(os.path.getmtime(s) for s in srcs)
    procs = []
    for part, defs in PARTS.items():
        so = _so(part)
        if force or not os.path.exists(so) or os.path.getmtime(so) < newest:
            procs.append(subprocess.Popen(
                ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O1", "-std=c++17", "-fPIC", "-shared",
                 "-fno-signed-zeros", "-ffinite-math-only", "--cuda-host-only", *defs, "-o", so, srcs[0]],
                stdout=subprocess.DEVNULL, stderr=subprocess.PIPE))
    for p in procs:
        err = p.communicate()[1]
        if p.returncode:
            raise RuntimeError("hostsim build failed:\n" + err.decode()[-3000:])
    return [_so(p) for p in PARTS]


def _lib_for(arm=None, law=False):
    """the part holding the row programs of this arm"""
    if law:
        part = "law"
    elif isinstance(arm, str):
        part = "static"
    else:
        n = arm["n_joints"]
        part = "rt13" if n <= 3 else "rt45" if n <= 5 else "rt6" if n == 6 else "rt7"
    if part not in _libs:
        if not _libs:
            build()  # once per process: rebuilds the parts that are older than hostsim.cpp or a kernel header
        _libs[part] = C.CDLL(_so(part))
    return _libs[part]


def _arm(arm):
    """arm: built-in name (static kernels) or a table dict (runtime-table kernels)"""
    if isinstance(arm, str):
        return arm.encode(), None, _abi.load_table(arm)["n_joints"]
    d = _abi.desc_from_table(arm)
    return None, C.byref(d), d.n_joints


def _in(a, dt):
    return None if a is None else np.ascontiguousarray(a, dtype=dt)


def _p(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


def dynamics(arm, q, dq=None, frame=None, x_off=None, want=("M",), dtype=np.float64):
    name, desc, n = _arm(arm)
    dt = np.dtype(dtype)
    q, dq = _in(q, dt), _in(dq, dt)
    B = q.shape[0]
    frame = 2 * n + 1 if frame is None else frame
    do = _abi.DynOut()
    res, bits = {}, 0
    for w in want:
        bits |= _WANT_BITS[w]
        res[w] = np.full((B,) + _OUT_SHAPES[w](n), np.nan, dt)
        setattr(do, w, res[w].ctypes.data)
    xo = None if x_off is None else (C.c_double * 3)(*[float(v) for v in x_off])
    rc = _lib_for(arm).hostsim_dynamics(name, desc, _dtype_code(dt), C.c_int64(B), _p(q), _p(dq), frame, xo,
                                C.c_uint32(bits), C.byref(do))
    assert rc == 0, rc
    return res


def osc_generate(arm, params, q, dq, target, target_velocity=None, integrated_error=None, u_null_ext=None,
                 training_signal=False, dtype=np.float64, handover=False):
    """handover=True (six task rows only): the two-pass form libabrk launches up to 262144 rows - first pass without
    the eigen-decomposition, deferred rows finished from their hand-over records; -> (..., rows deferred)"""
    name, desc, n = _arm(arm)
    dt = np.dtype(dtype)
    q, dq, target = _in(q, dt), _in(dq, dt), _in(target, dt)
    tv, une = _in(target_velocity, dt), _in(u_null_ext, dt)
    B = q.shape[0]
    u = np.full((B, n), np.nan, dt)
    ts = np.full((B, n), np.nan, dt) if training_signal else None
    if integrated_error is not None:
        assert integrated_error.dtype == dt and integrated_error.flags.c_contiguous
    if handover:
        nd = C.c_int64(0)
        rc = _lib_for(arm).hostsim_osc_handover(name, desc, _dtype_code(dt), C.byref(params), C.c_int64(B), _p(q),
                                                _p(dq), _p(target), _p(tv), _p(integrated_error), _p(une), _p(u),
                                                _p(ts), C.byref(nd))
        if rc == -1:  # not a six-row law: there is no second pass to emulate
            nd.value = -1
            rc = _lib_for(arm).hostsim_osc(name, desc, _dtype_code(dt), C.byref(params), C.c_int64(B), _p(q), _p(dq),
                                           _p(target), _p(tv), _p(integrated_error), _p(une), _p(u), _p(ts))
        assert rc == 0, rc
        return (u, ts, nd.value) if training_signal else (u, nd.value)
    rc = _lib_for(arm).hostsim_osc(name, desc, _dtype_code(dt), C.byref(params), C.c_int64(B), _p(q), _p(dq), _p(target),
                           _p(tv), _p(integrated_error), _p(une), _p(u), _p(ts))
    assert rc == 0, rc
    return (u, ts) if training_signal else u


def osc_generate_full(arm, params, q, dq, target, want=("Tx", "J", "M", "g"), target_velocity=None, u_null_ext=None,
                      dtype=np.float64):
    """the fused row program (osc_full_body): u, training_signal and the requested robot_config outputs"""
    name, desc, n = _arm(arm)
    dt = np.dtype(dtype)
    q, dq, target = _in(q, dt), _in(dq, dt), _in(target, dt)
    tv, une = _in(target_velocity, dt), _in(u_null_ext, dt)
    B = q.shape[0]
    u, ts = np.full((B, n), np.nan, dt), np.full((B, n), np.nan, dt)
    do, res, bits = _abi.DynOut(), {}, 0
    for w in want:
        res[w] = np.full((B,) + _OUT_SHAPES[w](n), np.nan, dt)
        setattr(do, w, res[w].ctypes.data)
        bits |= _WANT_BITS[w]
    rc = _lib_for(arm).hostsim_osc_full(name, desc, _dtype_code(dt), C.byref(params), C.c_int64(B), _p(q), _p(dq),
                                        _p(target), _p(tv), None, _p(une), _p(u), _p(ts), C.c_uint32(bits), C.byref(do))
    assert rc == 0, rc
    return u, ts, res


def sliding_generate(arm, params, q, dq, target, target_velocity=None, target_acc=None, want_s=False,
                     dtype=np.float64):
    name, desc, n = _arm(arm)
    dt = np.dtype(dtype)
    q, dq, target = _in(q, dt), _in(dq, dt), _in(target, dt)
    tv, ta = _in(target_velocity, dt), _in(target_acc, dt)
    B = q.shape[0]
    u = np.full((B, n), np.nan, dt)
    s = np.full((B, n), np.nan, dt)
    rc = _lib_for(arm).hostsim_sliding(name, desc, _dtype_code(dt), C.byref(params), C.c_int64(B), _p(q), _p(dq), _p(target),
                               _p(tv), _p(ta), _p(u), _p(s))
    assert rc == 0, rc
    return (u, s) if want_s else u


def joint_generate(arm, ctrl, account_for_gravity, q, dq, target=None, target_velocity=None, dtype=np.float64):
    name, desc, n = _arm(arm)
    dt = np.dtype(dtype)
    q, dq, target, tv = _in(q, dt), _in(dq, dt), _in(target, dt), _in(target_velocity, dt)
    B = q.shape[0]
    u = np.full((B, n), np.nan, dt)
    rc = _lib_for(arm).hostsim_joint(name, desc, _dtype_code(dt), C.byref(ctrl), int(bool(account_for_gravity)),
                             C.c_int64(B), _p(q), _p(dq), _p(target), _p(tv), _p(u))
    assert rc == 0, rc
    return u


def avoid_joint_limits_generate(n, params, q, u=None, dtype=np.float64):
    dt = np.dtype(dtype)
    q = _in(q, dt)
    B = q.shape[0]
    acc = u is not None
    u = np.full((B, n), np.nan, dt) if u is None else u
    rc = _lib_for(law=True).hostsim_limits(n, _dtype_code(dt), C.byref(params), C.c_int64(B), _p(q), _p(u), int(acc))
    assert rc == 0, rc
    return u


def floating_generate(arm, dynamic, task_space, q, dq=None, u=None, dtype=np.float64):
    name, desc, n = _arm(arm)
    dt = np.dtype(dtype)
    q, dq = _in(q, dt), _in(dq, dt)
    B = q.shape[0]
    acc = u is not None
    u = np.full((B, n), np.nan, dt) if u is None else u
    rc = _lib_for(arm).hostsim_floating(name, desc, _dtype_code(dt), int(bool(dynamic)), int(bool(task_space)),
                                        C.c_int64(B), _p(q), _p(dq), _p(u), int(acc))
    assert rc == 0, rc
    return u


def avoid_obstacles_generate(arm, params, q, u=None, dtype=np.float64, plain=False):
    """plain=False: what libabrk dispatches (orthogonal chains of three joints and more: phase A + one heavy pair at a
    time + finish, the row-level form of obstacles_lds_kernel); plain=True: the one-pass row program everywhere"""
    name, desc, n = _arm(arm)
    dt = np.dtype(dtype)
    q = _in(q, dt)
    B = q.shape[0]
    acc = u is not None
    u = np.full((B, n), np.nan, dt) if u is None else u
    rc = _lib_for(arm).hostsim_obstacles(name, desc, _dtype_code(dt), C.byref(params), C.c_int64(B), _p(q), _p(u),
                                         int(acc), int(plain))
    assert rc == 0, rc
    return u


def osc_law(n, params, J, M, dq, target, g=None, Cdq=None, xyz=None, R=None, q=None, target_velocity=None,
            integrated_error=None, u_null_ext=None, dtype=np.float64):
    dt = np.dtype(dtype)
    J, M, dq, target = _in(J, dt), _in(M, dt), _in(dq, dt), _in(target, dt)
    g, Cdq, xyz, R, q = _in(g, dt), _in(Cdq, dt), _in(xyz, dt), _in(R, dt), _in(q, dt)
    tv, une = _in(target_velocity, dt), _in(u_null_ext, dt)
    B = J.shape[0]
    u = np.full((B, n), np.nan, dt)
    ts = np.full((B, n), np.nan, dt)
    rc = _lib_for(law=True).hostsim_osc_law(n, _dtype_code(dt), C.byref(params), C.c_int64(B), _p(J), _p(M), _p(g), _p(Cdq), _p(xyz),
                               _p(R), _p(q), _p(dq), _p(target), _p(tv), _p(integrated_error), _p(une), _p(u), _p(ts))
    assert rc == 0, rc
    return u, ts


def osc_mx(n, M, J, threshold=1e-3, dtype=np.float64):
    dt = np.dtype(dtype)
    M, J = _in(M, dt), _in(J, dt)
    B, k = J.shape[0], J.shape[1]
    Mx, Minv = np.full((B, k, k), np.nan, dt), np.full((B, n, n), np.nan, dt)
    rc = _lib_for(law=True).hostsim_osc_mx(n, k, _dtype_code(dt), C.c_int64(B), _p(M), _p(J), C.c_double(threshold),
                                           _p(Mx), _p(Minv))
    assert rc == 0, rc
    return Mx, Minv


def sym6_eig(A, method):
    """the device code's 6 x 6 symmetric eigen-solvers (abrk_ctrl.h): method 0 `jacobi_eig`, 1 `ql_eig` -> (lam [B,6], V [B,6,6])"""
    A = _in(A, np.dtype(np.float64))
    B = A.shape[0]
    lam, V = np.full((B, 6), np.nan), np.full((B, 6, 6), np.nan)
    rc = _lib_for(law=True).hostsim_sym6_eig(int(method), C.c_int64(B), _p(A), _p(lam), _p(V))
    assert rc == 0, rc
    return lam, V


def osc6_tail(A, G, b):
    """the device code's truncating pseudo-inverse + tail of the six-row law (abrk_ctrl.h `osc6_tail`: Householder, the
    early-exit QL iteration, a tridiagonal solve): A [B,6,6], G [B,8,6], b [B,12] -> (u [B,6], ts [B,6], lexit [B], cut [B])"""
    A, G, b = (_in(x, np.dtype(np.float64)) for x in (A, G, b))
    B = A.shape[0]
    out, info = np.full((B, 12), np.nan), np.full((B, 2), np.nan)
    rc = _lib_for(law=True).hostsim_osc6_tail(C.c_int64(B), _p(A), _p(G), _p(b), _p(out), _p(info))
    assert rc == 0, rc
    return out[:, :6], out[:, 6:], info[:, 0].astype(int), info[:, 1]


def spd_inverse_small(A):
    """the device code's cofactor inverse of a symmetric 2 x 2 / 3 x 3 matrix (abrk_ctrl.h `spd_inverse_small`, the
    task-space inertia of the x,y,z / x,y law): A [B,K,K] -> (inv [B,K,K], det [B], ok [B] bool)"""
    A = _in(A, np.dtype(np.float64))
    B, K = A.shape[0], A.shape[1]
    inv, det, ok = np.full((B, K, K), np.nan), np.full(B, np.nan), np.zeros(B, np.int32)
    rc = _lib_for(law=True).hostsim_spd_inverse_small(int(K), C.c_int64(B), _p(A), _p(inv), _p(det), _p(ok))
    assert rc == 0, rc
    return inv, det, ok.astype(bool)


def sym3_eig(A, dtype=np.float64):
    """the device code's direct symmetric 3x3 eigen-solver (abrk_ctrl.h `sym3_eig`): -> (lam [B,3], V [B,3,3])"""
    dt = np.dtype(dtype)
    A = _in(A, dt)
    B = A.shape[0]
    lam, V = np.full((B, 3), np.nan, dt), np.full((B, 3, 3), np.nan, dt)
    rc = _lib_for(law=True).hostsim_sym3_eig(_dtype_code(dt), C.c_int64(B), _p(A), _p(lam), _p(V))
    assert rc == 0, rc
    return lam, V


def osc_velocity_limiting(params, u_task, dtype=np.float64):
    dt = np.dtype(dtype)
    u_task = _in(u_task, dt)
    out = np.full(u_task.shape, np.nan, dt)
    rc = _lib_for(law=True).hostsim_velocity_limiting(_dtype_code(dt), C.byref(params), C.c_int64(len(u_task)),
                                                      _p(u_task), _p(out))
    assert rc == 0, rc
    return out


def osc_orientation_forces(alg, R, abg, dtype=np.float64):
    dt = np.dtype(dtype)
    R, abg = _in(R, dt), _in(abg, dt)
    out = np.full((len(R), 3), np.nan, dt)
    rc = _lib_for(law=True).hostsim_orientation_forces(int(alg), _dtype_code(dt), C.c_int64(len(R)), _p(R), _p(abg),
                                                       _p(out))
    assert rc == 0, rc
    return out


def rollout_twolink(arm, params, plant, q0, dq0, target, n_steps, every, dtype=np.float64):
    name, desc, n = _arm(arm)
    dt = np.dtype(dtype)
    q, dq, target = _in(q0, dt).copy(), _in(dq0, dt).copy(), _in(target, dt)
    B = q.shape[0]
    n_chk = n_steps // every
    qt, dqt, ut = (np.full((B, n_chk, 2), np.nan, dt) for _ in range(3))
    rc = _lib_for(arm).hostsim_rollout(name, desc, _dtype_code(dt), C.byref(params), C.byref(plant), C.c_int64(B), n_steps, every,
                               _p(q), _p(dq), _p(target), _p(qt), _p(dqt), _p(ut))
    assert rc == 0, rc
    return q, dq, qt, dqt, ut


def ik_generate_path(arm, params, position, target, dtype=np.float64):
    name, desc, n = _arm(arm)
    dt = np.dtype(dtype)
    position, target = _in(position, dt), _in(target, dt)
    B, T = position.shape[0], int(params.n_timesteps)
    pp, vp = np.full((B, T, n), np.nan, dt), np.full((B, T, n), np.nan, dt)
    rc = _lib_for(arm).hostsim_ik(name, desc, _dtype_code(dt), C.byref(params), C.c_int64(B), _p(position), _p(target), _p(pp), _p(vp))
    assert rc == 0, rc
    return pp, vp
