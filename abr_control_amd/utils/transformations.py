"""The functions of abr_control/utils/transformations.py that the control path uses (euler_matrix
:973, quaternion_from_euler :1096, quaternion_from_matrix :1192, quaternion_multiply :1274, quaternion_conjugate
:1293, unit_vector :1632), evaluated by the kernels' own device functions through abrk_transformations_batch - one
value or a stack of B values per call.  Same names, argument order and result layout (quaternions are (w, x, y, z)).

Euler axes: 'rxyz' (what OSC uses, osc.py:164,178) and 'sxyz' (the reference's default; inverse_kinematics.py:73-82).
The other 22 axis sequences are not on the control path and raise ValueError here; the reference's module (plain
Python) remains the place for them."""
import numpy as np

from .. import engine

_AXES = {"rxyz": 0, "sxyz": 1}


def _rows(a, width):
    a = np.asarray(a, dtype=np.float64)
    single = a.ndim == 1
    a = np.ascontiguousarray(a.reshape(-1, width))
    return a, single


def quaternion_from_euler(ai, aj, ak, axes="sxyz"):
    """quaternion from Euler angles (transformations.py:1096-1150); scalars or arrays of B angles"""
    if axes not in _AXES:
        raise ValueError(f"axes {axes!r}: only {sorted(_AXES)} run on the GPU (the sequences the controllers use)")
    single = np.ndim(ai) == 0
    ang = np.stack(np.broadcast_arrays(np.atleast_1d(ai), np.atleast_1d(aj), np.atleast_1d(ak)), axis=-1)
    q = engine.transformations(_AXES[axes], np.ascontiguousarray(ang, dtype=np.float64))
    return q[0] if single else q


def euler_matrix(ai, aj, ak, axes="sxyz"):
    """homogeneous rotation matrix from Euler angles (transformations.py:973-1032); [4,4] or [B,4,4]"""
    if axes != "rxyz":
        raise ValueError(f"axes {axes!r}: only 'rxyz' runs on the GPU (osc.py:178)")
    single = np.ndim(ai) == 0
    ang = np.stack(np.broadcast_arrays(np.atleast_1d(ai), np.atleast_1d(aj), np.atleast_1d(ak)), axis=-1)
    R = engine.transformations(7, np.ascontiguousarray(ang, dtype=np.float64)).reshape(-1, 3, 3)
    M = np.zeros((len(R), 4, 4))
    M[:, :3, :3] = R
    M[:, 3, 3] = 1.0
    return M[0] if single else M


def quaternion_from_matrix(matrix, isprecise=False):
    """unit quaternion (w >= 0) of a rotation matrix [3,3], [4,4] or stacks of them (transformations.py:1192-1271).
    The matrix must be a rotation up to rounding (as robot_config.R is): the dominant eigenvector of K is taken by
    power iteration, which for a far-from-orthogonal matrix differs from the reference's eigh."""
    m = np.asarray(matrix, dtype=np.float64)
    single = m.ndim == 2
    m = m.reshape((-1,) + m.shape[-2:])[:, :3, :3]
    q = engine.transformations(2, np.ascontiguousarray(m).reshape(-1, 9))
    return q[0] if single else q


def quaternion_multiply(quaternion1, quaternion0):
    """product of two quaternions (transformations.py:1274-1290)"""
    q1, s1 = _rows(quaternion1, 4)
    q0, s0 = _rows(quaternion0, 4)
    B = max(len(q1), len(q0))
    q1, q0 = (np.ascontiguousarray(np.broadcast_to(q, (B, 4))) for q in (q1, q0))
    r = engine.transformations(3, q1, q0)
    return r[0] if (s1 and s0) else r


def quaternion_conjugate(quaternion):
    """conjugate (transformations.py:1293-1305)"""
    q, single = _rows(quaternion, 4)
    r = engine.transformations(4, q)
    return r[0] if single else r


def unit_vector(data, axis=None, out=None):
    """vectors of length 3 or 4 (positions, quaternions) normalised along the last axis (transformations.py:1632-1676)"""
    d = np.asarray(data, dtype=np.float64)
    if d.ndim == 0 or d.shape[-1] not in (3, 4) or axis not in (None, -1, d.ndim - 1) or (axis is None and d.ndim > 1):
        raise ValueError("unit_vector on the GPU handles 3- and 4-vectors along the last axis")
    r = engine.transformations(5 if d.shape[-1] == 4 else 6, np.ascontiguousarray(d.reshape(-1, d.shape[-1])))
    r = r.reshape(d.shape)
    if out is not None:
        out[...] = r
        return None
    return r
