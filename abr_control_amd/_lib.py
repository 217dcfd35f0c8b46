"""ctypes binding of libabrk.so (include/abrk.h).  No PyTorch, no fallback: if the shared
library is missing this module raises on first use, and every compute call raises when
there is no HIP device (ABRK_ENODEV)."""
import ctypes as C
import os

import numpy as np

from . import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ABRK_LIB_PATH") or os.path.join(_HERE, "libabrk.so")  # env override: kernel-variant experiments
_lib = None

_vp = C.c_void_p
_i64 = C.c_int64


class AbrkError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libabrk: {_abi.ERRORS.get(code, code)}: {msg}")
        self.code = code


def lib():
    """The loaded libabrk.so (built by `__graft_entry__.build()` / csrc/Makefile)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found - build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()' or make -C abr_control_amd/csrc). "
                "abr_control_amd has no CPU fallback."
            )
        L = C.CDLL(LIB_PATH)
        L.abrk_last_error.restype = C.c_char_p
        L.abrk_arm_builtin.argtypes = [C.c_char_p]
        L.abrk_arm_create.argtypes = [C.POINTER(_abi.ArmDesc)]
        L.abrk_arm_create_compiled.argtypes = [C.POINTER(_abi.ArmDesc), C.c_char_p]
        L.abrk_plugin_abi.restype = C.c_char_p
        L.abrk_arm_get_desc.argtypes = [C.c_int, C.POINTER(_abi.ArmDesc)]
        L.abrk_arm_destroy.argtypes = [C.c_int]
        L.abrk_dynamics_batch.argtypes = [
            C.c_int, C.c_int, _i64, _vp, _vp, C.c_int, C.POINTER(C.c_double), C.c_uint32,
            C.POINTER(_abi.DynOut), C.c_int, _vp]
        L.abrk_osc_generate_batch.argtypes = [
            C.c_int, C.c_int, C.POINTER(_abi.OSCParams), _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
            C.c_int, _vp]
        L.abrk_osc_generate_full_batch.argtypes = [
            C.c_int, C.c_int, C.POINTER(_abi.OSCParams), _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_uint32,
            C.POINTER(_abi.DynOut), C.c_int, _vp]
        L.abrk_osc_generate_coop_batch.argtypes = [
            C.c_int, C.c_int, C.POINTER(_abi.OSCParams), _i64, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp]
        L.abrk_osc_generate_sharded.argtypes = [
            C.c_int, C.c_int, C.POINTER(_abi.OSCParams), _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int,
            C.POINTER(C.c_int)]
        L.abrk_sliding_generate_sharded.argtypes = [
            C.c_int, C.c_int, C.POINTER(_abi.SlidingParams), _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int,
            C.POINTER(C.c_int)]
        L.abrk_joint_generate_sharded.argtypes = [
            C.c_int, C.c_int, C.POINTER(_abi.NullCtrl), C.c_int, _i64, _vp, _vp, _vp, _vp, _vp, C.c_int,
            C.POINTER(C.c_int)]
        L.abrk_dynamics_sharded.argtypes = [
            C.c_int, C.c_int, _i64, _vp, _vp, C.c_int, C.POINTER(C.c_double), C.c_uint32, C.POINTER(_abi.DynOut),
            C.c_int, C.POINTER(C.c_int)]
        _pp, _pi64, _pi32 = C.POINTER(_vp), C.POINTER(_i64), C.POINTER(C.c_int32)
        L.abrk_shard_stream.restype = _vp
        L.abrk_shard_stream.argtypes = [C.c_int, C.c_int]
        L.abrk_osc_generate_resident.argtypes = [C.c_int, C.c_int, C.POINTER(_abi.OSCParams),
                                                 C.POINTER(_abi.ShardCut)] + [_pp] * 8
        L.abrk_sliding_generate_resident.argtypes = [C.c_int, C.c_int, C.POINTER(_abi.SlidingParams),
                                                     C.POINTER(_abi.ShardCut)] + [_pp] * 7
        L.abrk_joint_generate_resident.argtypes = [C.c_int, C.c_int, C.POINTER(_abi.NullCtrl), C.c_int,
                                                   C.POINTER(_abi.ShardCut)] + [_pp] * 5
        L.abrk_dynamics_resident.argtypes = [C.c_int, C.c_int, C.POINTER(_abi.ShardCut), _pp, _pp, C.c_int,
                                             C.POINTER(C.c_double), C.c_uint32, C.POINTER(_abi.DynOut)]
        L.abrk_shards_sync.argtypes = [C.POINTER(_abi.ShardCut)]
        L.abrk_plans_launch.argtypes = [C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int]
        L.abrk_osc_law_batch.argtypes = [C.c_int, C.c_int, C.POINTER(_abi.OSCParams), _i64] + [_vp] * 14 + [C.c_int, _vp]
        L.abrk_osc_mx_batch.argtypes = [C.c_int, C.c_int, C.c_int, _i64, _vp, _vp, C.c_double, _vp, _vp, C.c_int, _vp]
        L.abrk_osc_velocity_limiting_batch.argtypes = [C.c_int, C.POINTER(_abi.OSCParams), _i64, _vp, _vp, C.c_int, _vp]
        L.abrk_osc_orientation_forces_batch.argtypes = [C.c_int, C.c_int, _i64, _vp, _vp, _vp, C.c_int, _vp]
        L.abrk_transformations_batch.argtypes = [C.c_int, C.c_int, _i64, _vp, _vp, _vp, C.c_int, _vp]
        L.abrk_osc_plan_create.argtypes = L.abrk_osc_generate_batch.argtypes
        L.abrk_sliding_plan_create.argtypes = [
            C.c_int, C.c_int, C.POINTER(_abi.SlidingParams), _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, _vp]
        L.abrk_plan_begin.argtypes = [C.c_int, _vp]
        L.abrk_plan_launch.argtypes = [C.c_int]
        L.abrk_plan_launch_graph.argtypes = [C.c_int, C.c_int]
        L.abrk_plan_launch_repeat.argtypes = [C.c_int, C.c_int]
        L.abrk_plan_destroy.argtypes = [C.c_int]
        L.abrk_ik_generate_path_batch.argtypes = [C.c_int, C.c_int, C.POINTER(_abi.IkParams), _i64, _vp, _vp, _vp, _vp,
                                                  C.c_int, _vp]
        L.abrk_twolink_step_batch.argtypes = [C.c_int, C.POINTER(_abi.TwoLinkPlant), _i64, _vp, _vp, _vp, C.c_int, _vp]
        L.abrk_osc_rollout_twolink_batch.argtypes = [
            C.c_int, C.c_int, C.POINTER(_abi.OSCParams), C.POINTER(_abi.TwoLinkPlant), _i64, C.c_int32, C.c_int32,
            _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, _vp]
        L.abrk_sliding_generate_batch.argtypes = [
            C.c_int, C.c_int, C.POINTER(_abi.SlidingParams), _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
            C.c_int, _vp]
        L.abrk_joint_generate_batch.argtypes = [
            C.c_int, C.c_int, C.POINTER(_abi.NullCtrl), C.c_int, _i64, _vp, _vp, _vp, _vp, _vp, C.c_int, _vp]
        L.abrk_avoid_joint_limits_generate_batch.argtypes = [
            C.c_int, C.c_int, C.POINTER(_abi.LimitsParams), _i64, _vp, _vp, C.c_int, C.c_int, _vp]
        L.abrk_floating_generate_batch.argtypes = [
            C.c_int, C.c_int, C.c_int, C.c_int, _i64, _vp, _vp, _vp, C.c_int, C.c_int, _vp]
        L.abrk_avoid_obstacles_generate_batch.argtypes = [
            C.c_int, C.c_int, C.POINTER(_abi.ObstaclesParams), _i64, _vp, _vp, C.c_int, C.c_int, _vp]
        L.abrk_device_name.argtypes = [C.c_int, C.c_char_p, C.c_size_t]
        L.abrk_malloc.restype = _vp
        L.abrk_malloc.argtypes = [C.c_int, C.c_size_t]
        L.abrk_free.argtypes = [C.c_int, _vp]
        L.abrk_memcpy_h2d.argtypes = [C.c_int, _vp, _vp, C.c_size_t, _vp]
        L.abrk_memcpy_d2h.argtypes = [C.c_int, _vp, _vp, C.c_size_t, _vp]
        L.abrk_memset.argtypes = [C.c_int, _vp, C.c_int, C.c_size_t, _vp]
        L.abrk_stream_create.restype = _vp
        L.abrk_stream_create.argtypes = [C.c_int]
        L.abrk_stream_destroy.argtypes = [C.c_int, _vp]
        L.abrk_stream_sync.argtypes = [C.c_int, _vp]
        L.abrk_scratch_stats.argtypes = [C.c_int, C.POINTER(_abi.ScratchInfo)]
        L.abrk_device_sync.argtypes = [C.c_int]
        L.abrk_event_create.restype = _vp
        L.abrk_event_create.argtypes = [C.c_int]
        L.abrk_event_destroy.argtypes = [C.c_int, _vp]
        L.abrk_event_record.argtypes = [C.c_int, _vp, _vp]
        L.abrk_event_elapsed_ms.argtypes = [C.c_int, _vp, _vp, C.POINTER(C.c_float)]
        _lib = L
    return _lib


class SingularMatrixError(AbrkError, np.linalg.LinAlgError):
    """ABRK_ESINGULAR: a row's joint-space inertia matrix is not positive definite.  Also a numpy.linalg.LinAlgError -
    what the reference's `np.linalg.inv(M)` raises at controllers/osc.py:136 - so `except LinAlgError` keeps working."""


def check(rc):
    if rc == _abi.ESINGULAR:
        raise SingularMatrixError(rc, lib().abrk_last_error().decode())
    if rc < 0:
        raise AbrkError(rc, lib().abrk_last_error().decode())
    return rc


def device_count():
    return lib().abrk_device_count()


def device_name(device=0):
    buf = C.create_string_buffer(256)
    check(lib().abrk_device_name(device, buf, 256))
    return buf.value.decode()


def scratch_stats(device=0):
    """the library's own device scratch (abrk_scratch_stats): a dict of its counters"""
    info = _abi.ScratchInfo()
    check(lib().abrk_scratch_stats(device, C.byref(info)))
    return {k: int(getattr(info, k)) for k, _ in _abi.ScratchInfo._fields_}


NP_DTYPE = {_abi.F64: np.float64, _abi.F32: np.float32}


class DeviceArray:
    """A caller-owned device buffer (hipMalloc through the C ABI): shape + dtype + pointer.
    Passing DeviceArrays to the batched calls keeps them zero-copy and asynchronous."""

    def __init__(self, shape, dtype=np.float64, device=0):
        self.shape = tuple(int(s) for s in np.atleast_1d(shape))
        self.dtype = np.dtype(dtype)
        self.device = device
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        self.ptr = lib().abrk_malloc(device, self.nbytes)
        if not self.ptr:
            raise AbrkError(-3, lib().abrk_last_error().decode())

    @classmethod
    def from_numpy(cls, a, device=0, stream=None):
        a = np.ascontiguousarray(a)
        d = cls(a.shape, a.dtype, device)
        check(lib().abrk_memcpy_h2d(device, d.ptr, a.ctypes.data, a.nbytes, stream))
        return d

    def numpy(self, stream=None):
        """copy to a new host array on `stream` (a Stream or a raw handle; None = the NULL stream) and wait for it.  The
        copy drains that stream, so a singular batch enqueued on it earlier raises here (SingularMatrixError)"""
        out = np.empty(self.shape, self.dtype)
        check(lib().abrk_memcpy_d2h(self.device, out.ctypes.data, self.ptr, self.nbytes, getattr(stream, "ptr", stream)))
        return out

    def zero_(self, stream=None):
        check(lib().abrk_memset(self.device, self.ptr, 0, self.nbytes, getattr(stream, "ptr", stream)))
        return self

    def rows(self, lo, hi):
        """rows [lo, hi) of this array as a DeviceArray of their own: a VIEW (same memory, no copy; keeps this array
        alive) - how several control loops share one batch buffer (engine.MergedLoops)"""
        lo, hi = int(lo), int(hi)
        if not 0 <= lo <= hi <= self.shape[0]:
            raise IndexError(f"rows [{lo}, {hi}) outside 0..{self.shape[0]}")
        v = DeviceArray.__new__(DeviceArray)
        v.shape, v.dtype, v.device = (hi - lo,) + self.shape[1:], self.dtype, self.device
        row_bytes = (self.nbytes // self.shape[0]) if self.shape[0] else 0
        v.nbytes = (hi - lo) * row_bytes
        v.ptr = (self.ptr or 0) + lo * row_bytes
        v._base = self  # a view frees nothing
        return v

    def copy_from_numpy(self, a, stream=None):
        """overwrite the buffer with a host array of the same shape (synchronous)"""
        a = np.ascontiguousarray(a, dtype=self.dtype)
        if a.shape != self.shape:
            raise ValueError(f"expected {self.shape}, got {a.shape}")
        if self.nbytes:
            check(lib().abrk_memcpy_h2d(self.device, self.ptr, a.ctypes.data, a.nbytes, getattr(stream, "ptr", stream)))
        return self

    def free(self):
        if getattr(self, "_base", None) is not None:  # a view (rows()): the memory belongs to its base array
            self.ptr = None
            return
        if self.ptr:
            lib().abrk_free(self.device, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Stream:
    def __init__(self, device=0):
        self.device = device
        self.ptr = lib().abrk_stream_create(device)
        if not self.ptr:
            raise AbrkError(-2, lib().abrk_last_error().decode())

    def sync(self):
        check(lib().abrk_stream_sync(self.device, self.ptr))

    def __del__(self):
        try:
            if self.ptr:
                lib().abrk_stream_destroy(self.device, self.ptr)
        except Exception:
            pass


class BorrowedStream(Stream):
    """a stream handle this object does not own (the library's per-(device, slot) shard streams, abrk_shard_stream):
    usable wherever a Stream is, never destroyed from here"""

    def __init__(self, device, ptr):
        self.device, self.ptr = device, ptr

    def __del__(self):
        pass


def shard_stream(device, slot):
    """the library's own stream of the `slot`-th shard on `device` (what the *_resident entry points use when the caller
    names no streams)"""
    p = lib().abrk_shard_stream(int(device), int(slot))
    if not p:
        raise AbrkError(-2, lib().abrk_last_error().decode())
    return BorrowedStream(int(device), p)


class Event:
    def __init__(self, device=0):
        self.device = device
        self.ptr = lib().abrk_event_create(device)
        if not self.ptr:
            raise AbrkError(-2, lib().abrk_last_error().decode())

    def record(self, stream=None):
        check(lib().abrk_event_record(self.device, self.ptr, stream.ptr if stream else None))

    def elapsed_ms_since(self, start):
        ms = C.c_float()
        check(lib().abrk_event_elapsed_ms(self.device, start.ptr, self.ptr, C.byref(ms)))
        return ms.value

    def __del__(self):
        try:
            if self.ptr:
                lib().abrk_event_destroy(self.device, self.ptr)
        except Exception:
            pass
