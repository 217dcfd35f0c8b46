"""Batched calls into libabrk.so.

Every function takes either NumPy arrays (staged through device scratch by the library,
results returned as NumPy) or `DeviceArray`s (zero-copy, asynchronous on `stream`, results
returned as `DeviceArray`s).  Shapes are `[B, ...]`, row-major; dtype float64 or float32
selects the arithmetic of the kernels.
"""
import ctypes as C

import numpy as np

from . import _abi
from ._lib import NP_DTYPE, AbrkError, DeviceArray, check, lib  # noqa: F401

_OUT_SHAPES = {
    "Tx": lambda n: (3,),
    "J": lambda n: (6, n),
    "M": lambda n: (n, n),
    "g": lambda n: (n,),
    "C": lambda n: (n, n),
    "dJ": lambda n: (6, n),
    "R": lambda n: (3, 3),
    "T": lambda n: (4, 4),
    "Tinv": lambda n: (4, 4),
    "quat": lambda n: (4,),
}
_WANT_BITS = {
    "Tx": _abi.WANT_TX, "J": _abi.WANT_J, "M": _abi.WANT_M, "g": _abi.WANT_G, "C": _abi.WANT_C,
    "dJ": _abi.WANT_DJ, "R": _abi.WANT_R, "T": _abi.WANT_T, "Tinv": _abi.WANT_TINV, "quat": _abi.WANT_QUAT,
}


def _dtype_code(dtype):
    dt = np.dtype(dtype)
    if dt == np.float64:
        return _abi.F64
    if dt == np.float32:
        return _abi.F32
    raise TypeError(f"abr_control_amd computes in float64 or float32, not {dt}")


class _Args:
    """Marshals array arguments: keeps host copies alive, decides host/device mode."""

    def __init__(self, dtype):
        self.np_dtype = np.dtype(dtype)
        self.code = _dtype_code(dtype)
        self.keep = []
        self.on_device = None

    def _mode(self, dev):
        if self.on_device is None:
            self.on_device = dev
        elif self.on_device != dev:
            raise TypeError("mixing DeviceArray and NumPy arguments in one call is not supported")

    def inp(self, a, shape, name):
        if a is None:
            return None
        if isinstance(a, DeviceArray):
            self._mode(True)
            if a.dtype != self.np_dtype or a.shape != tuple(shape):
                raise ValueError(f"{name}: expected {self.np_dtype}{tuple(shape)}, got {a.dtype}{a.shape}")
            return a.ptr
        self._mode(False)
        h = np.ascontiguousarray(a, dtype=self.np_dtype)
        if h.shape != tuple(shape):
            raise ValueError(f"{name}: expected shape {tuple(shape)}, got {h.shape}")
        self.keep.append(h)
        return h.ctypes.data

    def out(self, given, shape, device, name):
        """-> (pointer, object to return)"""
        if self.on_device:
            if given is None:
                given = DeviceArray(shape, self.np_dtype, device)
            elif not isinstance(given, DeviceArray) or given.shape != tuple(shape) or given.dtype != self.np_dtype:
                raise ValueError(f"{name}: output must be a DeviceArray {self.np_dtype}{tuple(shape)}")
            return given.ptr, given
        if given is None:
            given = np.empty(shape, self.np_dtype)
        elif (not isinstance(given, np.ndarray) or given.shape != tuple(shape) or given.dtype != self.np_dtype
              or not given.flags.c_contiguous):
            raise ValueError(f"{name}: output must be a C-contiguous ndarray {self.np_dtype}{tuple(shape)}")
        return given.ctypes.data, given


def _sp(stream):
    return None if stream is None else getattr(stream, "ptr", stream)


def dynamics(arm_id, n, q, dq=None, frame=None, x_off=None, want=("M",), dtype=np.float64, device=0,
             stream=None, out=None):
    """robot_config.{Tx,J,M,g,C,dJ,R,T,T_inv,quaternion} for a batch (one launch, shared FK).
    Returns {name: array[B, ...]} for every name in `want`."""
    a = _Args(dtype)
    B = q.shape[0]
    frame = 2 * n + 1 if frame is None else frame
    qp = a.inp(q, (B, n), "q")
    dqp = a.inp(dq, (B, n), "dq")
    bits = 0
    do = _abi.DynOut()
    res = {}
    for name in want:
        bits |= _WANT_BITS[name]
        ptr, obj = a.out(None if out is None else out.get(name), (B,) + _OUT_SHAPES[name](n), device, name)
        setattr(do, name, ptr)
        res[name] = obj
    xo = None
    if x_off is not None:
        xo = (C.c_double * 3)(*[float(v) for v in x_off])
    check(lib().abrk_dynamics_batch(arm_id, a.code, B, qp, dqp, frame, xo, bits, C.byref(do), device, _sp(stream)))
    return res


def osc_generate(arm_id, n, params, q, dq, target, target_velocity=None, integrated_error=None,
                 u_null_ext=None, u=None, training_signal=False, dtype=np.float64, device=0, stream=None,
                 want=None, out=None):
    """OSC.generate for a batch.  `integrated_error` ([B,6]) is updated in place when ki != 0.
    Returns u, or (u, training_signal) when training_signal is True / an output array.
    want: subset of ("Tx", "J", "M", "g", "C", "dJ") -> the fused kernel (abrk_osc_generate_full_batch) also writes those
    robot_config outputs of the controller's ref_frame / xyz_offset; the return value gains a dict of them
    (`out`: optional dict of preallocated arrays)."""
    a = _Args(dtype)
    B = q.shape[0]
    qp = a.inp(q, (B, n), "q")
    dqp = a.inp(dq, (B, n), "dq")
    tp = a.inp(target, (B, 6), "target")
    tvp = a.inp(target_velocity, (B, 6), "target_velocity")
    unp = a.inp(u_null_ext, (B, n), "u_null_ext")
    iep = None
    if integrated_error is not None:
        if isinstance(integrated_error, DeviceArray):
            iep = a.inp(integrated_error, (B, 6), "integrated_error")
        else:
            if (not isinstance(integrated_error, np.ndarray) or integrated_error.dtype != a.np_dtype
                    or integrated_error.shape != (B, 6) or not integrated_error.flags.c_contiguous):
                raise ValueError("integrated_error must be a C-contiguous ndarray [B,6] of the call dtype")
            a._mode(False)
            iep = integrated_error.ctypes.data
    up, uo = a.out(u, (B, n), device, "u")
    tsp, tso = None, None
    if training_signal is not False and training_signal is not None:
        tsp, tso = a.out(None if training_signal is True else training_signal, (B, n), device, "training_signal")
    if want:
        bits, do, res = 0, _abi.DynOut(), {}
        for name in want:
            if name not in ("Tx", "J", "M", "g", "C", "dJ"):
                raise ValueError(f"the fused kernel offers Tx, J, M, g, C, dJ - not {name!r}")
            bits |= _WANT_BITS[name]
            ptr, obj = a.out(None if out is None else out.get(name), (B,) + _OUT_SHAPES[name](n), device, name)
            setattr(do, name, ptr)
            res[name] = obj
        check(lib().abrk_osc_generate_full_batch(arm_id, a.code, C.byref(params), B, qp, dqp, tp, tvp, iep, unp, up,
                                                 tsp, bits, C.byref(do), device, _sp(stream)))
        return (uo, tso, res) if tso is not None else (uo, res)
    check(lib().abrk_osc_generate_batch(arm_id, a.code, C.byref(params), B, qp, dqp, tp, tvp, iep, unp, up, tsp,
                                        device, _sp(stream)))
    return (uo, tso) if tso is not None else uo


def osc_generate_coop(arm_id, n, params, q, dq, target, lanes_per_arm, u=None, training_signal=False, dtype=np.float64,
                      device=0, stream=None):
    """The wave-cooperative mapping of the plain OSC law (abrk_osc_generate_coop_batch; ur5, fp64): K = lanes_per_arm
    lanes per arm instance.  Measurement variant - see profiles/round2/coop_ab.md."""
    a = _Args(dtype)
    B = q.shape[0]
    qp = a.inp(q, (B, n), "q")
    dqp = a.inp(dq, (B, n), "dq")
    tp = a.inp(target, (B, 6), "target")
    up, uo = a.out(u, (B, n), device, "u")
    tsp, tso = None, None
    if training_signal is not False and training_signal is not None:
        tsp, tso = a.out(None if training_signal is True else training_signal, (B, n), device, "training_signal")
    check(lib().abrk_osc_generate_coop_batch(arm_id, a.code, C.byref(params), B, qp, dqp, tp, up, tsp,
                                             int(lanes_per_arm), device, _sp(stream)))
    return (uo, tso) if tso is not None else uo


def osc_generate_sharded(arm_id, n, params, q, dq, target, devices, target_velocity=None, integrated_error=None,
                         u_null_ext=None, u=None, training_signal=False, dtype=np.float64):
    """OSC.generate of ONE host batch over several devices (abrk_osc_generate_sharded): contiguous row shards, shard g
    on devices[g], no collective.  NumPy arrays only.  Returns u or (u, training_signal)."""
    a = _Args(dtype)
    B = q.shape[0]
    for name, arr in (("q", q), ("dq", dq), ("target", target), ("target_velocity", target_velocity),
                      ("u_null_ext", u_null_ext), ("u", u)):
        if isinstance(arr, DeviceArray):
            raise TypeError(f"{name}: the sharded call takes NumPy arrays (a DeviceArray lives on one device)")
    qp = a.inp(q, (B, n), "q")
    dqp = a.inp(dq, (B, n), "dq")
    tp = a.inp(target, (B, 6), "target")
    tvp = a.inp(target_velocity, (B, 6), "target_velocity")
    unp = a.inp(u_null_ext, (B, n), "u_null_ext")
    iep = None
    if integrated_error is not None:
        if (not isinstance(integrated_error, np.ndarray) or integrated_error.dtype != a.np_dtype
                or integrated_error.shape != (B, 6) or not integrated_error.flags.c_contiguous):
            raise ValueError("integrated_error must be a C-contiguous ndarray [B,6] of the call dtype")
        iep = integrated_error.ctypes.data
    up, uo = a.out(u, (B, n), 0, "u")
    tsp, tso = None, None
    if training_signal is not False and training_signal is not None:
        tsp, tso = a.out(None if training_signal is True else training_signal, (B, n), 0, "training_signal")
    devs = (C.c_int * len(devices))(*[int(d) for d in devices])
    check(lib().abrk_osc_generate_sharded(arm_id, a.code, C.byref(params), B, qp, dqp, tp, tvp, iep, unp, up, tsp,
                                          len(devices), devs))
    return (uo, tso) if tso is not None else uo


def _host_only(**arrays):
    for name, arr in arrays.items():
        if isinstance(arr, DeviceArray):
            raise TypeError(f"{name}: the sharded calls take NumPy arrays (a DeviceArray lives on one device)")


def sliding_generate_sharded(arm_id, n, params, q, dq, target, devices, target_velocity=None, target_acc=None,
                             want_s=False, dtype=np.float64):
    """Sliding.generate of ONE host batch over several devices (abrk_sliding_generate_sharded)"""
    _host_only(q=q, dq=dq, target=target, target_velocity=target_velocity, target_acc=target_acc)
    a = _Args(dtype)
    B = q.shape[0]
    nt = 3 if params.cartesian else n
    qp, dqp = a.inp(q, (B, n), "q"), a.inp(dq, (B, n), "dq")
    tp = a.inp(target, (B, nt), "target")
    tvp, tap = a.inp(target_velocity, (B, nt), "target_velocity"), a.inp(target_acc, (B, nt), "target_acc")
    up, uo = a.out(None, (B, n), 0, "u")
    sp, so = a.out(None, (B, n), 0, "s") if want_s else (None, None)
    devs = (C.c_int * len(devices))(*[int(d) for d in devices])
    check(lib().abrk_sliding_generate_sharded(arm_id, a.code, C.byref(params), B, qp, dqp, tp, tvp, tap, up, sp,
                                              len(devices), devs))
    return (uo, so) if want_s else uo


def joint_generate_sharded(arm_id, n, ctrl, account_for_gravity, q, dq, devices, target=None, target_velocity=None,
                           dtype=np.float64):
    """Joint / Damping / RestingConfig.generate of ONE host batch over several devices (abrk_joint_generate_sharded)"""
    _host_only(q=q, dq=dq, target=target, target_velocity=target_velocity)
    a = _Args(dtype)
    B = q.shape[0]
    qp, dqp = a.inp(q, (B, n), "q"), a.inp(dq, (B, n), "dq")
    tp, tvp = a.inp(target, (B, n), "target"), a.inp(target_velocity, (B, n), "target_velocity")
    up, uo = a.out(None, (B, n), 0, "u")
    devs = (C.c_int * len(devices))(*[int(d) for d in devices])
    check(lib().abrk_joint_generate_sharded(arm_id, a.code, C.byref(ctrl), int(bool(account_for_gravity)), B, qp, dqp, tp,
                                            tvp, up, len(devices), devs))
    return uo


def dynamics_sharded(arm_id, n, q, devices, dq=None, frame=None, x_off=None, want=("M",), dtype=np.float64):
    """robot_config.{Tx,J,M,g,C,dJ,R,T,T_inv,quaternion} of ONE host batch over several devices (abrk_dynamics_sharded)"""
    _host_only(q=q, dq=dq)
    a = _Args(dtype)
    B = q.shape[0]
    frame = 2 * n + 1 if frame is None else frame
    qp, dqp = a.inp(q, (B, n), "q"), a.inp(dq, (B, n), "dq")
    bits, do, res = 0, _abi.DynOut(), {}
    for name in want:
        bits |= _WANT_BITS[name]
        ptr, obj = a.out(None, (B,) + _OUT_SHAPES[name](n), 0, name)
        setattr(do, name, ptr)
        res[name] = obj
    xo = None if x_off is None else (C.c_double * 3)(*[float(v) for v in x_off])
    devs = (C.c_int * len(devices))(*[int(d) for d in devices])
    check(lib().abrk_dynamics_sharded(arm_id, a.code, B, qp, dqp, frame, xo, bits, C.byref(do), len(devices), devs))
    return res


# ---------------------------------------------------------------------------- resident shards
class _Cut:
    """abrk_shard_cut + the pointer tables of one resident call (sharding.ShardedArray arguments)"""

    def __init__(self, like, dtype, streams=None):
        self.devices, self.rows = list(like.devices), list(like.rows)
        self.G = len(self.devices)
        self.np_dtype, self.code = np.dtype(dtype), _dtype_code(dtype)
        self._dev = (C.c_int32 * self.G)(*self.devices)
        self._rows = (C.c_int64 * self.G)(*self.rows)
        self._streams = None
        if streams is not None:
            if len(streams) != self.G:
                raise ValueError(f"{len(streams)} streams for {self.G} shards")
            self._streams = (C.c_void_p * self.G)(*[_sp(st) for st in streams])
        self.c = _abi.ShardCut(self.G, self._dev, self._rows, self._streams)

    def tab(self, arr, tail, name):
        """pointer table of a ShardedArray (None -> NULL), checked against the cut"""
        if arr is None:
            return None
        if list(arr.devices) != self.devices or list(arr.rows) != self.rows:
            raise ValueError(f"{name}: cut over other devices / rows than q")
        if arr.dtype != self.np_dtype or tuple(arr.shape[1:]) != tuple(tail):
            raise ValueError(f"{name}: expected {self.np_dtype}[B{tuple(tail)}], got {arr.dtype}{arr.shape}")
        return (C.c_void_p * self.G)(*[p.ptr for p in arr.parts])

    def out(self, given, tail, name):
        from .sharding import ShardedArray

        if given is None:
            given = ShardedArray.empty((sum(self.rows),) + tuple(tail), self.np_dtype, self.devices, rows=self.rows)
        return self.tab(given, tail, name), given


def osc_generate_resident(arm_id, n, params, q, dq, target, target_velocity=None, integrated_error=None,
                          u_null_ext=None, u=None, training_signal=False, dtype=np.float64, streams=None):
    """OSC.generate on a batch whose shards LIVE on the devices (abrk_osc_generate_resident): every array a
    sharding.ShardedArray cut the same way; the call only enqueues (shard g on streams[g], or on the library's own
    stream of its (device, slot)).  integrated_error [B,6] stays with its shards.  Returns u, or (u, training_signal),
    as ShardedArrays - not yet complete: sync with `shards_sync(u)` / MultiDevice.sync()."""
    c = _Cut(q, dtype, streams)
    qp, dqp, tp = c.tab(q, (n,), "q"), c.tab(dq, (n,), "dq"), c.tab(target, (6,), "target")
    tvp, unp = c.tab(target_velocity, (6,), "target_velocity"), c.tab(u_null_ext, (n,), "u_null_ext")
    iep = c.tab(integrated_error, (6,), "integrated_error")
    up, uo = c.out(u, (n,), "u")
    tsp, tso = None, None
    if training_signal is not False and training_signal is not None:
        tsp, tso = c.out(None if training_signal is True else training_signal, (n,), "training_signal")
    check(lib().abrk_osc_generate_resident(arm_id, c.code, C.byref(params), C.byref(c.c), qp, dqp, tp, tvp, iep, unp, up,
                                           tsp))
    return (uo, tso) if tso is not None else uo


def sliding_generate_resident(arm_id, n, params, q, dq, target, target_velocity=None, target_acc=None, u=None,
                              want_s=False, dtype=np.float64, streams=None):
    """Sliding.generate on resident shards (abrk_sliding_generate_resident)"""
    c = _Cut(q, dtype, streams)
    nt = 3 if params.cartesian else n
    qp, dqp, tp = c.tab(q, (n,), "q"), c.tab(dq, (n,), "dq"), c.tab(target, (nt,), "target")
    tvp, tap = c.tab(target_velocity, (nt,), "target_velocity"), c.tab(target_acc, (nt,), "target_acc")
    up, uo = c.out(u, (n,), "u")
    sp, so = c.out(None if want_s is True else want_s, (n,), "s") if want_s else (None, None)
    check(lib().abrk_sliding_generate_resident(arm_id, c.code, C.byref(params), C.byref(c.c), qp, dqp, tp, tvp, tap, up, sp))
    return (uo, so) if want_s else uo


def joint_generate_resident(arm_id, n, ctrl, account_for_gravity, q, dq, target=None, target_velocity=None, u=None,
                            dtype=np.float64, streams=None):
    """Joint / Damping / RestingConfig.generate on resident shards (abrk_joint_generate_resident)"""
    c = _Cut(q, dtype, streams)
    qp, dqp = c.tab(q, (n,), "q"), c.tab(dq, (n,), "dq")
    tp, tvp = c.tab(target, (n,), "target"), c.tab(target_velocity, (n,), "target_velocity")
    up, uo = c.out(u, (n,), "u")
    check(lib().abrk_joint_generate_resident(arm_id, c.code, C.byref(ctrl), int(bool(account_for_gravity)), C.byref(c.c),
                                             qp, dqp, tp, tvp, up))
    return uo


def dynamics_resident(arm_id, n, q, dq=None, frame=None, x_off=None, want=("M",), dtype=np.float64, streams=None,
                      out=None):
    """robot_config.{Tx,J,M,g,C,dJ,R,T,T_inv,quaternion} on resident shards (abrk_dynamics_resident):
    {name: ShardedArray}"""
    c = _Cut(q, dtype, streams)
    frame = 2 * n + 1 if frame is None else frame
    qp, dqp = c.tab(q, (n,), "q"), c.tab(dq, (n,), "dq")
    bits, res = 0, {}
    dos = (_abi.DynOut * c.G)()
    for name in want:
        bits |= _WANT_BITS[name]
        _, obj = c.out(None if out is None else out.get(name), _OUT_SHAPES[name](n), name)
        for g in range(c.G):
            setattr(dos[g], name, obj.parts[g].ptr)
        res[name] = obj
    xo = None if x_off is None else (C.c_double * 3)(*[float(v) for v in x_off])
    check(lib().abrk_dynamics_resident(arm_id, c.code, C.byref(c.c), qp, dqp, frame, xo, bits, dos))
    return res


def shards_sync(like, streams=None):
    """drain the streams of a resident batch (abrk_shards_sync): raises SingularMatrixError once if any shard met a
    singular M"""
    c = _Cut(like, like.dtype, streams)
    check(lib().abrk_shards_sync(C.byref(c.c)))


def plans_launch(plans, repeat=1, graph=False):
    """`repeat` ticks of several recorded plans (one per shard / device) from ONE call (abrk_plans_launch)"""
    ids = (C.c_int * len(plans))(*[p.id for p in plans])
    check(lib().abrk_plans_launch(ids, len(plans), int(repeat), 1 if graph else 0))


def osc_law(n, params, J, M, dq, target, g=None, Cdq=None, xyz=None, R=None, q=None, target_velocity=None,
            integrated_error=None, u_null_ext=None, training_signal=False, dtype=np.float64, device=0, stream=None):
    """The OSC control law on caller-supplied dynamics (abrk_osc_law_batch): J [B,6,n], M [B,n,n] and,
    as the controller options require, g [B,n], Cdq [B,n], xyz [B,3], R [B,3,3], q [B,n]."""
    a = _Args(dtype)
    B = J.shape[0]
    Jp = a.inp(J, (B, 6, n), "J")
    Mp = a.inp(M, (B, n, n), "M")
    gp = a.inp(g, (B, n), "g")
    cp = a.inp(Cdq, (B, n), "Cdq")
    xp = a.inp(xyz, (B, 3), "xyz")
    Rp = a.inp(R, (B, 3, 3), "R")
    qp = a.inp(q, (B, n), "q")
    dqp = a.inp(dq, (B, n), "dq")
    tp = a.inp(target, (B, 6), "target")
    tvp = a.inp(target_velocity, (B, 6), "target_velocity")
    unp = a.inp(u_null_ext, (B, n), "u_null_ext")
    iep = None
    if integrated_error is not None:
        if isinstance(integrated_error, DeviceArray):
            iep = a.inp(integrated_error, (B, 6), "integrated_error")
        else:
            if (not isinstance(integrated_error, np.ndarray) or integrated_error.dtype != a.np_dtype
                    or integrated_error.shape != (B, 6) or not integrated_error.flags.c_contiguous):
                raise ValueError("integrated_error must be a C-contiguous ndarray [B,6] of the call dtype")
            iep = integrated_error.ctypes.data
    up, uo = a.out(None, (B, n), device, "u")
    tsp, tso = (None, None)
    if training_signal:
        tsp, tso = a.out(None, (B, n), device, "training_signal")
    check(lib().abrk_osc_law_batch(n, a.code, C.byref(params), B, Jp, Mp, gp, cp, xp, Rp, qp, dqp, tp, tvp, iep, unp,
                                   up, tsp, device, _sp(stream)))
    return (uo, tso) if training_signal else uo


def osc_mx(n, M, J, threshold=1e-3, want_minv=True, dtype=np.float64, device=0, stream=None):
    """OSC._Mx (osc.py:120-147) for B rows: M [B,n,n], J [B,k,n] -> (Mx [B,k,k], M_inv [B,n,n] or None)"""
    a = _Args(dtype)
    B, k = J.shape[0], J.shape[1]
    Mp = a.inp(M, (B, n, n), "M")
    Jp = a.inp(J, (B, k, n), "J")
    xp, xo = a.out(None, (B, k, k), device, "Mx")
    ip, io = a.out(None, (B, n, n), device, "M_inv") if want_minv else (None, None)
    check(lib().abrk_osc_mx_batch(n, k, a.code, B, Mp, Jp, float(threshold), xp, ip, device, _sp(stream)))
    return xo, io


def osc_velocity_limiting(params, u_task, dtype=np.float64, device=0, stream=None):
    """OSC._velocity_limiting (osc.py:198-215) for B rows of u_task [B,6]"""
    a = _Args(dtype)
    B = u_task.shape[0]
    ip = a.inp(u_task, (B, 6), "u_task")
    op, oo = a.out(None, (B, 6), device, "out")
    check(lib().abrk_osc_velocity_limiting_batch(a.code, C.byref(params), B, ip, op, device, _sp(stream)))
    return oo


def osc_orientation_forces(algorithm, R, target_abg, dtype=np.float64, device=0, stream=None):
    """OSC._calc_orientation_forces (osc.py:149-196) from R [B,3,3] = robot_config.R(ref_frame, q), target_abg [B,3]"""
    a = _Args(dtype)
    B = R.shape[0]
    Rp = a.inp(R, (B, 3, 3), "R")
    tp = a.inp(target_abg, (B, 3), "target_abg")
    op, oo = a.out(None, (B, 3), device, "u_task_orientation")
    check(lib().abrk_osc_orientation_forces_batch(int(algorithm), a.code, B, Rp, tp, op, device, _sp(stream)))
    return oo


_TF_WIDTHS = {0: (3, 0, 4), 1: (3, 0, 4), 2: (9, 0, 4), 3: (4, 4, 4), 4: (4, 0, 4), 5: (4, 0, 4), 6: (3, 0, 3),
              7: (3, 0, 9)}


def transformations(op, a, b=None, dtype=np.float64, device=0, stream=None):
    """abrk_transformations_batch: op = ABRK_TF_* (include/abrk.h); a [B,wa], b [B,wb] or None -> out [B,wo]"""
    wa, wb, wo = _TF_WIDTHS[int(op)]
    ar = _Args(dtype)
    B = a.shape[0]
    ap = ar.inp(a, (B, wa), "a")
    bp = ar.inp(b, (B, wb), "b") if wb else None
    op_, oo = ar.out(None, (B, wo), device, "out")
    check(lib().abrk_transformations_batch(int(op), ar.code, B, ap, bp, op_, device, _sp(stream)))
    return oo


def sliding_generate(arm_id, n, params, q, dq, target, target_velocity=None, target_acc=None, u=None,
                     want_s=False, dtype=np.float64, device=0, stream=None):
    a = _Args(dtype)
    B = q.shape[0]
    nt = 3 if params.cartesian else n
    qp = a.inp(q, (B, n), "q")
    dqp = a.inp(dq, (B, n), "dq")
    tp = a.inp(target, (B, nt), "target")
    tvp = a.inp(target_velocity, (B, nt), "target_velocity")
    tap = a.inp(target_acc, (B, nt), "target_acc")
    up, uo = a.out(u, (B, n), device, "u")
    sp, so = (None, None)
    if want_s:
        sp, so = a.out(None, (B, n), device, "s")
    check(lib().abrk_sliding_generate_batch(arm_id, a.code, C.byref(params), B, qp, dqp, tp, tvp, tap, up, sp,
                                            device, _sp(stream)))
    return (uo, so) if want_s else uo


def joint_generate(arm_id, n, ctrl, account_for_gravity, q, dq, target=None, target_velocity=None, u=None,
                   dtype=np.float64, device=0, stream=None):
    a = _Args(dtype)
    B = q.shape[0]
    qp = a.inp(q, (B, n), "q")
    dqp = a.inp(dq, (B, n), "dq")
    tp = a.inp(target, (B, n), "target")
    tvp = a.inp(target_velocity, (B, n), "target_velocity")
    up, uo = a.out(u, (B, n), device, "u")
    check(lib().abrk_joint_generate_batch(arm_id, a.code, C.byref(ctrl), int(bool(account_for_gravity)), B, qp, dqp,
                                          tp, tvp, up, device, _sp(stream)))
    return uo


def avoid_joint_limits_generate(n, params, q, u=None, accumulate=False, dtype=np.float64, device=0, stream=None):
    """AvoidJointLimits.generate for B states; accumulate: u += signal (u must then be given)."""
    a = _Args(dtype)
    B = q.shape[0]
    qp = a.inp(q, (B, n), "q")
    up, uo = a.out(u, (B, n), device, "u")
    check(lib().abrk_avoid_joint_limits_generate_batch(n, a.code, C.byref(params), B, qp, up,
                                                       int(bool(accumulate and u is not None)), device, _sp(stream)))
    return uo


def floating_generate(arm_id, n, dynamic, task_space, q, dq=None, u=None, accumulate=False, dtype=np.float64,
                      device=0, stream=None):
    """Floating.generate for B states."""
    a = _Args(dtype)
    B = q.shape[0]
    qp = a.inp(q, (B, n), "q")
    dqp = a.inp(dq, (B, n), "dq") if dynamic else None
    up, uo = a.out(u, (B, n), device, "u")
    check(lib().abrk_floating_generate_batch(arm_id, a.code, int(bool(dynamic)), int(bool(task_space)), B, qp, dqp,
                                             up, int(bool(accumulate and u is not None)), device, _sp(stream)))
    return uo


def avoid_obstacles_generate(arm_id, n, params, q, u=None, accumulate=False, dtype=np.float64, device=0,
                             stream=None):
    """AvoidObstacles.generate for B states (obstacles shared by all rows)."""
    a = _Args(dtype)
    B = q.shape[0]
    qp = a.inp(q, (B, n), "q")
    up, uo = a.out(u, (B, n), device, "u")
    check(lib().abrk_avoid_obstacles_generate_batch(arm_id, a.code, C.byref(params), B, qp, up,
                                                    int(bool(accumulate and u is not None)), device, _sp(stream)))
    return uo


def _inout(a, arr, shape, name):
    """in/out state array: DeviceArray, or a C-contiguous ndarray of the call dtype updated in place"""
    if isinstance(arr, DeviceArray):
        return a.inp(arr, shape, name)
    if (not isinstance(arr, np.ndarray) or arr.dtype != a.np_dtype or arr.shape != tuple(shape)
            or not arr.flags.c_contiguous):
        raise ValueError(f"{name} must be a C-contiguous ndarray {tuple(shape)} of the call dtype (updated in place)")
    a._mode(False)
    return arr.ctypes.data


def twolink_step(plant, q, dq, u, dtype=np.float64, device=0, stream=None):
    """ArmSim._step for a batch: (q, dq) [B,2] advanced in place by torques u [B,2]."""
    a = _Args(dtype)
    B = q.shape[0]
    qp, dqp = _inout(a, q, (B, 2), "q"), _inout(a, dq, (B, 2), "dq")
    up = a.inp(u, (B, 2), "u")
    check(lib().abrk_twolink_step_batch(a.code, C.byref(plant), B, qp, dqp, up, device, _sp(stream)))


def osc_rollout_twolink(arm_id, params, plant, q, dq, target, n_steps, every=0, integrated_error=None,
                        want_traj=False, dtype=np.float64, device=0, stream=None):
    """n_steps x { OSC.generate ; ArmSim._step } in one launch.  q, dq [B,2] are advanced in place.
    Returns (q_traj, dq_traj, u_traj) [B, n_steps // every, 2] when want_traj, else None."""
    a = _Args(dtype)
    B = q.shape[0]
    qp, dqp = _inout(a, q, (B, 2), "q"), _inout(a, dq, (B, 2), "dq")
    tp = a.inp(target, (B, 6), "target")
    iep = None if integrated_error is None else _inout(a, integrated_error, (B, 6), "integrated_error")
    outs, ptrs = (None, None, None), (None, None, None)
    if want_traj:
        n_chk = n_steps // every if every else 0
        pairs = [a.out(None, (B, n_chk, 2), device, nm) for nm in ("q_traj", "dq_traj", "u_traj")]
        ptrs, outs = tuple(p for p, _ in pairs), tuple(o for _, o in pairs)
    check(lib().abrk_osc_rollout_twolink_batch(arm_id, a.code, C.byref(params), C.byref(plant), B, int(n_steps),
                                               int(every), qp, dqp, tp, iep, ptrs[0], ptrs[1], ptrs[2], device,
                                               _sp(stream)))
    return outs if want_traj else None


class Plan:
    """One control tick recorded as a launch plan (abrk_plan_begin .. abrk_plan_end): every engine call made inside
    the `with` block - on DeviceArrays, with this plan's device and stream - is validated and converted once and its
    kernel launch kept; `launch()` then only enqueues the recorded kernels, `launch_graph(repeat)` replays `repeat`
    ticks as one hipGraph launch.

        with engine.Plan(device, stream) as tick:
            engine.avoid_joint_limits_generate(n, lim, q, u=une, dtype=dt, device=device, stream=stream)
            engine.avoid_obstacles_generate(arm, n, obs, q, u=une, accumulate=True, ..., stream=stream)
            engine.osc_generate(arm, n, params, q, dq, target, u_null_ext=une, u=u, ..., stream=stream)
        tick.launch()

    The arrays must stay alive as long as the plan (results the engine allocates are returned to the caller as
    usual; `keep()` parks references on the plan)."""

    def __init__(self, device=0, stream=None):
        self.device, self.stream, self.id = device, stream, None
        self._keep = [stream]
        self._launch = lib().abrk_plan_launch

    def __enter__(self):
        check(lib().abrk_plan_begin(self.device, _sp(self.stream)))
        return self

    def __exit__(self, exc_type, exc, tb):
        if exc_type is not None:
            lib().abrk_plan_abort()
            return False
        self.id = check(lib().abrk_plan_end())
        return False

    def keep(self, *objs):
        self._keep.extend(objs)
        return self

    def launch(self):
        rc = self._launch(self.id)
        if rc < 0:
            check(rc)

    def launch_graph(self, repeat):
        """`repeat` consecutive launches as one hipGraph launch (needs a plan on an explicit stream)"""
        check(lib().abrk_plan_launch_graph(self.id, int(repeat)))

    def launch_repeat(self, repeat):
        """`repeat` consecutive plain launches enqueued by one C call (abrk_plan_launch_repeat)"""
        check(lib().abrk_plan_launch_repeat(self.id, int(repeat)))

    def close(self):
        pid, self.id = self.id, None
        if pid is not None:
            lib().abrk_plan_destroy(pid)

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass


class OscPlan(Plan):
    """A validated, pre-converted OSC launch on fixed device buffers (abrk_osc_plan_create): `launch()`
    only enqueues the kernel on the plan's stream - the per-tick cost of a control loop whose state lives
    on the GPU.  All arrays must be DeviceArrays; update their contents between launches."""

    def __init__(self, arm_id, n, params, q, dq, target, u, target_velocity=None, integrated_error=None,
                 u_null_ext=None, training_signal=None, dtype=np.float64, device=0, stream=None):
        Plan.__init__(self, device, stream)
        a = _Args(dtype)
        B = q.shape[0]
        arrs = dict(q=(q, (B, n)), dq=(dq, (B, n)), target=(target, (B, 6)), target_velocity=(target_velocity, (B, 6)),
                    integrated_error=(integrated_error, (B, 6)), u_null_ext=(u_null_ext, (B, n)), u=(u, (B, n)),
                    training_signal=(training_signal, (B, n)))
        ptr = {}
        for name, (arr, shape) in arrs.items():
            if arr is not None and not isinstance(arr, DeviceArray):
                raise TypeError(f"{name}: launch plans take DeviceArrays")
            ptr[name] = a.inp(arr, shape, name)
        self._keep += [v[0] for v in arrs.values()] + [params]
        self.id = check(lib().abrk_osc_plan_create(
            arm_id, a.code, C.byref(params), B, ptr["q"], ptr["dq"], ptr["target"], ptr["target_velocity"],
            ptr["integrated_error"], ptr["u_null_ext"], ptr["u"], ptr["training_signal"], device, _sp(stream)))


class SlidingPlan(Plan):
    """The same for Sliding.generate (abrk_sliding_plan_create): launch(), launch_graph(repeat)"""

    def __init__(self, arm_id, n, params, q, dq, target, u, target_velocity=None, target_acc=None, s=None,
                 dtype=np.float64, device=0, stream=None):
        Plan.__init__(self, device, stream)
        a = _Args(dtype)
        B = q.shape[0]
        nt = 3 if params.cartesian else n
        arrs = dict(q=(q, (B, n)), dq=(dq, (B, n)), target=(target, (B, nt)), target_velocity=(target_velocity, (B, nt)),
                    target_acc=(target_acc, (B, nt)), u=(u, (B, n)), s=(s, (B, n)))
        ptr = {}
        for name, (arr, shape) in arrs.items():
            if arr is not None and not isinstance(arr, DeviceArray):
                raise TypeError(f"{name}: launch plans take DeviceArrays")
            ptr[name] = a.inp(arr, shape, name)
        self._keep += [v[0] for v in arrs.values()] + [params]
        self.id = check(lib().abrk_sliding_plan_create(
            arm_id, a.code, C.byref(params), B, ptr["q"], ptr["dq"], ptr["target"], ptr["target_velocity"],
            ptr["target_acc"], ptr["u"], ptr["s"], device, _sp(stream)))


class MergedLoops:
    """Several INDEPENDENT control loops that run the same controller on one GPU, evaluated by ONE launch per tick.

    Why: independent loops on streams of their own do not overlap beyond two streams - the runtime maps streams onto a
    few hardware queues and the command processor serialises the dependent kernel chains that share one (measured,
    profiles/round6/concurrent_streams.md: 16 streams of 4096-row graph replays reach 3.0 G evaluations/s, while the
    same 65 536 rows as one launch take 5.4 us: 12.2 G/s; more hardware queues - GPU_MAX_HW_QUEUES=8 / 16 - make it
    4x WORSE).  A config-sized step occupies 64 of the chip's 1024 SIMDs, so the remedy is to put the loops' rows into
    one batch: loop i owns rows [lo_i, hi_i) of shared q / dq / target / u buffers (`loop(i)`: DeviceArray VIEWS - each
    loop writes its states and reads its torques without knowing of the others), and `launch()` / `launch_graph(K)`
    evaluate every loop's rows together.  The loops share the controller's parameters (one `abrk_osc_params` per
    launch); loops with different gains need launches of their own.

        loops = engine.MergedLoops(rc.arm_id, rc.N_JOINTS, params, [4096] * 16, device=0)
        loops.loop(3).q.copy_from_numpy(q3) ...        # every loop feeds its own rows
        loops.launch(); loops.stream.sync()
        u3 = loops.loop(3).u.numpy(loops.stream)"""

    def __init__(self, arm_id, n, params, rows_per_loop, dtype=np.float64, device=0, stream=None, target_velocity=False,
                 training_signal=False):
        from ._lib import Stream

        self.rows = [int(r) for r in rows_per_loop]
        if not self.rows or min(self.rows) < 1:
            raise ValueError("every loop needs at least one row")
        self.bounds = np.concatenate([[0], np.cumsum(self.rows)]).tolist()
        B = self.bounds[-1]
        self.device, self.stream = device, stream if stream is not None else Stream(device)
        mk = lambda w: DeviceArray((B, w), dtype, device)
        self.q, self.dq, self.target, self.u = mk(n), mk(n), mk(6), mk(n)
        self.target_velocity = mk(6).zero_() if target_velocity else None
        self.integrated_error = mk(6).zero_() if params.ki != 0 else None
        self.training_signal = mk(n) if training_signal else None
        with Plan(device, self.stream) as self.plan:
            osc_generate(arm_id, n, params, self.q, self.dq, self.target, self.target_velocity, self.integrated_error,
                         None, self.u, self.training_signal if training_signal else False, dtype=dtype, device=device,
                         stream=self.stream)

    def loop(self, i):
        """loop i's rows of every buffer, as DeviceArray views: .q .dq .target .u (.target_velocity .integrated_error
        .training_signal where present)"""
        import types

        lo, hi = self.bounds[i], self.bounds[i + 1]
        v = lambda a: None if a is None else a.rows(lo, hi)
        return types.SimpleNamespace(q=v(self.q), dq=v(self.dq), target=v(self.target), u=v(self.u),
                                     target_velocity=v(self.target_velocity), integrated_error=v(self.integrated_error),
                                     training_signal=v(self.training_signal), rows=(lo, hi))

    def launch(self):
        self.plan.launch()

    def launch_graph(self, repeat):
        self.plan.launch_graph(repeat)

    def close(self):
        self.plan.close()


def ik_generate_path(arm_id, n, params, position, target, dtype=np.float64, device=0, stream=None,
                     position_path=None, velocity_path=None):
    """InverseKinematics.generate_path for B paths: position [B,n], target [B,6] (xyz + Euler 'sxyz').
    Returns (position_path, velocity_path), each [B, n_timesteps, n]."""
    a = _Args(dtype)
    B = position.shape[0]
    T = int(params.n_timesteps)
    qp = a.inp(position, (B, n), "position")
    tp = a.inp(target, (B, 6), "target")
    ppp, ppo = a.out(position_path, (B, T, n), device, "position_path")
    vpp, vpo = a.out(velocity_path, (B, T, n), device, "velocity_path")
    check(lib().abrk_ik_generate_path_batch(arm_id, a.code, C.byref(params), B, qp, tp, ppp, vpp, device, _sp(stream)))
    return ppo, vpo
