"""ctypes front-end of oracle/libabrk_oracle.so (the plain-C restatement of the reference's
hot path).  TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg - never by the abr_control_amd package."""
import ctypes as C
import os
import subprocess

import numpy as np

from abr_control_amd import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libabrk_oracle.so")
_lib = None
_dp = C.POINTER(C.c_double)


def build(force=False):
    src = os.path.join(_HERE, "abrk_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s"], check=True,
                       stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _c(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


class Oracle:
    """Per-arm oracle.  All methods take ONE state (like the reference) unless named *_batch."""

    def __init__(self, table):
        self.table = table
        self.desc = _abi.desc_from_table(table)
        self.n = self.desc.n_joints
        self._d = C.byref(self.desc)
        self.L = lib()

    def fid(self, name):
        return _abi.frame_id(name, self.n)

    def _x(self, x):
        return _c(np.zeros(3) if x is None else x)

    def T(self, name, q):
        out = np.zeros((4, 4))
        assert self.L.abrk_oracle_T(self._d, self.fid(name), _p(_c(q)), _p(out)) == 0
        return out

    def Tx(self, name, q, x=None):
        out = np.zeros(3)
        assert self.L.abrk_oracle_Tx(self._d, self.fid(name), _p(_c(q)), _p(self._x(x)), _p(out)) == 0
        return out

    def J(self, name, q, x=None):
        out = np.zeros((6, self.n))
        assert self.L.abrk_oracle_J(self._d, self.fid(name), _p(_c(q)), _p(self._x(x)), _p(out)) == 0
        return out

    def dJ(self, name, q, dq, x=None):
        out = np.zeros((6, self.n))
        assert self.L.abrk_oracle_dJ(self._d, self.fid(name), _p(_c(q)), _p(_c(dq)), _p(self._x(x)), _p(out)) == 0
        return out

    def M(self, q):
        out = np.zeros((self.n, self.n))
        self.L.abrk_oracle_M(self._d, _p(_c(q)), _p(out))
        return out

    def g(self, q):
        out = np.zeros(self.n)
        self.L.abrk_oracle_g(self._d, _p(_c(q)), _p(out))
        return out

    def C(self, q, dq):
        out = np.zeros((self.n, self.n))
        self.L.abrk_oracle_C(self._d, _p(_c(q)), _p(_c(dq)), _p(out))
        return out

    def R(self, name, q):
        out = np.zeros((3, 3))
        assert self.L.abrk_oracle_R(self._d, self.fid(name), _p(_c(q)), _p(out)) == 0
        return out

    def T_inv(self, name, q):
        out = np.zeros((4, 4))
        assert self.L.abrk_oracle_Tinv(self._d, self.fid(name), _p(_c(q)), _p(out)) == 0
        return out

    def quaternion(self, name, q):
        out = np.zeros(4)
        assert self.L.abrk_oracle_quaternion(self._d, self.fid(name), _p(_c(q)), _p(out)) == 0
        return out

    # ---- controllers, batched drivers (row loop in C) ----
    def osc_batch(self, params, q, dq, target, target_velocity=None, integrated_error=None,
                  u_null_ext=None, want_training=False):
        q, dq, target = _c(q), _c(dq), _c(target)
        B = q.shape[0]
        tv, une = _c(target_velocity), _c(u_null_ext)
        u = np.zeros((B, self.n))
        ts = np.zeros((B, self.n)) if want_training else None
        rc = self.L.abrk_oracle_osc_generate_batch(
            self._d, C.byref(params), C.c_int64(B), _p(q), _p(dq), _p(target), _p(tv),
            _p(integrated_error), _p(une), _p(u), _p(ts))
        assert rc == 0, rc
        return (u, ts) if want_training else u

    def sliding_batch(self, params, q, dq, target, target_velocity=None, target_acc=None):
        q, dq, target = _c(q), _c(dq), _c(target)
        B = q.shape[0]
        u = np.zeros((B, self.n))
        s = np.zeros((B, self.n))
        rc = self.L.abrk_oracle_sliding_generate_batch(
            self._d, C.byref(params), C.c_int64(B), _p(q), _p(dq), _p(target),
            _p(_c(target_velocity)), _p(_c(target_acc)), _p(u), _p(s))
        assert rc == 0, rc
        return u, s

    def joint_batch(self, ctrl, account_for_gravity, q, dq, target=None, target_velocity=None):
        q, dq = _c(q), _c(dq)
        B = q.shape[0]
        target, tv = _c(target), _c(target_velocity)
        u = np.zeros((B, self.n))
        for b in range(B):
            rc = self.L.abrk_oracle_joint_generate(
                self._d, C.byref(ctrl), int(account_for_gravity), _p(q[b]), _p(dq[b]),
                None if target is None else _p(target[b]),
                None if tv is None else _p(tv[b]), _p(u[b]))
            assert rc == 0
        return u


    def floating_batch(self, dynamic, task_space, q, dq=None):
        """Floating.generate per row -> u [B,n], diag [B,2] (det, s_min/s_max of Mx_inv)"""
        q = _c(q)
        B = q.shape[0]
        dq = np.zeros_like(q) if dq is None else _c(dq)
        u, diag = np.zeros((B, self.n)), np.zeros((B, 2))
        for b in range(B):
            rc = self.L.abrk_oracle_floating_generate(self._d, int(dynamic), int(task_space), _p(q[b]), _p(dq[b]),
                                                      _p(u[b]), _p(diag[b]))
            assert rc == 0
        return u, diag

    def avoid_obstacles_batch(self, params, q):
        """AvoidObstacles.generate per row -> u [B,n], diag [B,2] (distance from the pinv threshold, smallest
        relative mobility of a closest point)"""
        q = _c(q)
        B = q.shape[0]
        u, diag = np.zeros((B, self.n)), np.zeros((B, 2))
        for b in range(B):
            rc = self.L.abrk_oracle_avoid_obstacles_generate(self._d, C.byref(params), _p(q[b]), _p(u[b]), _p(diag[b]))
            assert rc == 0
        return u, diag


def avoid_joint_limits_batch(n, params, q):
    """AvoidJointLimits.generate per row"""
    q = _c(q)
    u = np.zeros_like(q)
    for b in range(q.shape[0]):
        rc = lib().abrk_oracle_avoid_joint_limits_generate(int(n), C.byref(params), _p(q[b]), _p(u[b]))
        assert rc == 0
    return u


def rollout_twolink(table, params, plant, q0, dq0, target, n_steps, every):
    """closed loop OSC.generate + ArmSim._step (examples/PyGame/force_osc_xy.py:57-78) on the oracle"""
    o = Oracle(table)
    q, dq, target = _c(q0).copy(), _c(dq0).copy(), _c(target)
    B = q.shape[0]
    n_chk = n_steps // every
    qt, dqt, ut = np.zeros((B, n_chk, 2)), np.zeros((B, n_chk, 2)), np.zeros((B, n_chk, 2))
    rc = o.L.abrk_oracle_rollout_twolink(o._d, C.byref(params), C.byref(plant), C.c_int64(B), n_steps, every,
                                         _p(q), _p(dq), _p(target), None, _p(qt), _p(dqt), _p(ut))
    assert rc == 0, rc
    return q, dq, qt, dqt, ut


def ik_paths(table, params, position, target):
    """InverseKinematics.generate_path on the oracle, one path per row"""
    o = Oracle(table)
    position, target = _c(position), _c(target)
    B, T = position.shape[0], int(params.n_timesteps)
    pp, vp = np.zeros((B, T, o.n)), np.zeros((B, T, o.n))
    for b in range(B):
        rc = o.L.abrk_oracle_ik_generate_path(o._d, C.byref(params), _p(position[b]), _p(target[b]), _p(pp[b]), _p(vp[b]))
        assert rc == 0
    return pp, vp


def quat_from_matrix(R):
    out = np.zeros(4)
    lib().abrk_oracle_quat_from_matrix(_p(_c(R)), _p(out))
    return out


def quat_from_euler_rxyz(a, b, c):
    out = np.zeros(4)
    lib().abrk_oracle_quat_from_euler_rxyz(C.c_double(a), C.c_double(b), C.c_double(c), _p(out))
    return out


def euler_matrix_rxyz(a, b, c):
    out = np.zeros((3, 3))
    lib().abrk_oracle_euler_matrix_rxyz(C.c_double(a), C.c_double(b), C.c_double(c), _p(out))
    return out


def quat_mul(q1, q0):
    out = np.zeros(4)
    lib().abrk_oracle_quat_mul(_p(_c(q1)), _p(_c(q0)), _p(out))
    return out


def osc_mx(M, J, threshold=1e-3):
    """OSC._Mx (osc.py:120-147) for one (M [n,n], J [k,n]) -> (Mx [k,k], M_inv [n,n])"""
    M, J = _c(M), _c(J)
    n, k = M.shape[0], J.shape[0]
    Mx, Minv = np.zeros((k, k)), np.zeros((n, n))
    lib().abrk_oracle_osc_mx(int(n), int(k), _p(M), _p(J), C.c_double(threshold), _p(Mx), _p(Minv))
    return Mx, Minv


def osc_orientation_forces(algorithm, R_e, target_abg):
    """OSC._calc_orientation_forces (osc.py:149-196) from R_e = robot_config.R(ref_frame, q)"""
    out = np.zeros(3)
    rc = lib().abrk_oracle_osc_orientation_forces(int(algorithm), _p(_c(R_e)), _p(_c(target_abg)), _p(out))
    assert rc == 0, rc
    return out


def osc_velocity_limiting(params, u_task):
    """OSC._velocity_limiting (osc.py:198-215)"""
    out = _c(u_task).copy()
    lib().abrk_oracle_osc_velocity_limiting(C.byref(params), _p(out))
    return out
