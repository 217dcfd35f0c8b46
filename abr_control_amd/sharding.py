"""Batch sharding across GPUs.  Rows are independent, so a shard is a contiguous row range and NO
collective sits on the data path.  Two ways to own several GPUs:
  * one process per GPU (bench.py --gpus N): what the ranks exchange - a barrier and the maximum of their timings -
    goes through HostGroup below: a few bytes over a local socket, no communication library;
  * ONE process, every device (MultiDevice): host batches through the *_sharded entry points, or - the form a control
    loop wants - shards that LIVE on the devices (ShardedArray) through the *_resident entry points: a call only
    enqueues, results stay in per-device buffers until .numpy() gathers them, recorded plans replay K ticks on every
    device from one call."""
import json
import os
import socket
import struct
import time


def dist_env():
    """(rank, local_rank, world_size) as a launcher (`python -m torch.distributed.run`, mpirun wrappers, a shell
    loop) exports them: RANK, LOCAL_RANK, WORLD_SIZE"""
    return (int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)),
            int(os.environ.get("WORLD_SIZE", 1)))


class HostGroup:
    """Barrier / max / gather between the ranks of ONE node (bench.py --gpus N: one process per GPU): rank 0 listens on
    a Unix socket in the abstract namespace (Linux), keyed by a per-run token when the launcher exports one
    (ABRK_GROUP_KEY - bench.py's self-launch does - or TORCHELASTIC_RUN_ID), by MASTER_PORT and by the launcher's pid
    (the ranks are siblings - the launcher's own rendezvous store may occupy MASTER_PORT itself, so no TCP port is
    taken); every operation is one exchange of a small JSON value: each rank sends its value, rank 0 answers with the
    list of all of them."""

    def __init__(self, rank, world, timeout=120.0, key=None):
        if world < 1 or not 0 <= rank < world:
            raise ValueError(f"rank {rank} outside world of {world}")
        self.rank, self.world = rank, world
        run = os.environ.get("ABRK_GROUP_KEY") or os.environ.get("TORCHELASTIC_RUN_ID") or ""
        key = key or f"{run}_{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}"
        name = "\0abrk_hostgroup_" + key
        if rank == 0:
            self.srv = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
            try:
                self.srv.bind(name)
            except OSError as e:
                self.srv.close()
                raise RuntimeError(f"HostGroup: the rendezvous name for key {key!r} is taken ({e}): another run of the "
                                   f"same launcher is using it - export a distinct ABRK_GROUP_KEY per run") from e
            self.srv.listen(world)
            self.srv.settimeout(timeout)
            self.peers = [None] * world
            for _ in range(world - 1):
                c, _addr = self.srv.accept()
                c.settimeout(timeout)
                r = struct.unpack("<i", self._recvn(c, 4))[0]
                if not 0 < r < world or self.peers[r] is not None:
                    c.close()
                    raise RuntimeError(f"HostGroup: a peer announced rank {r} (world {world}"
                                       f"{', already connected' if 0 < r < world else ''}): stray or duplicate rank")
                self.peers[r] = c
        else:
            deadline = time.monotonic() + timeout
            while True:
                self.sock = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
                try:
                    self.sock.connect(name)
                    break
                except OSError:  # rank 0 is not up yet
                    self.sock.close()
                    if time.monotonic() > deadline:
                        raise
                    time.sleep(0.02)
            self.sock.settimeout(timeout)
            self.sock.sendall(struct.pack("<i", rank))

    @staticmethod
    def _recvn(c, n):
        buf = b""
        while len(buf) < n:
            part = c.recv(n - len(buf))
            if not part:
                raise ConnectionError("a rank of the host group went away")
            buf += part
        return buf

    @classmethod
    def _send(cls, c, obj):
        raw = json.dumps(obj).encode()
        c.sendall(struct.pack("<q", len(raw)) + raw)

    @classmethod
    def _recv(cls, c):
        return json.loads(cls._recvn(c, struct.unpack("<q", cls._recvn(c, 8))[0]))

    def exchange(self, value=None):
        """-> [value of rank 0, ..., value of rank world-1] on every rank; returns once every rank has called it"""
        if self.world == 1:
            return [value]
        if self.rank == 0:
            vals = [value] + [self._recv(self.peers[r]) for r in range(1, self.world)]
            for r in range(1, self.world):
                self._send(self.peers[r], vals)
            return vals
        self._send(self.sock, value)
        return self._recv(self.sock)

    def barrier(self):
        self.exchange(None)

    def max(self, x):
        return max(self.exchange(float(x)))

    def close(self):
        self.barrier()
        if self.world > 1:
            for c in (self.peers[1:] + [self.srv]) if self.rank == 0 else [self.sock]:
                c.close()


def shard_range(B, rank, world):
    """contiguous rows [lo, hi) of rank `rank`: sizes differ by at most one, cover [0, B) exactly"""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world of {world}")
    base, extra = divmod(B, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_rows(arrays, rank, world):
    """slice every [B, ...] array (None passes through) to this rank's rows"""
    B = next(a.shape[0] for a in arrays if a is not None)
    lo, hi = shard_range(B, rank, world)
    return [None if a is None else a[lo:hi] for a in arrays]


class ShardedArray:
    """A [B, ...] array cut into contiguous row shards that LIVE on devices: shard g = rows shard_range(B, g, G) on
    devices[g] (a device may be named several times), each a DeviceArray.  `from_numpy` scatters once, `.numpy()`
    gathers; in between the *_resident calls (engine.osc_generate_resident, MultiDevice.generate) read and write the
    shards in place - SURVEY.md 8e: "results remain in per-device buffers unless the caller asks for host arrays"."""

    def __init__(self, parts, devices):
        if not parts or len(parts) != len(devices):
            raise ValueError("one DeviceArray per shard")
        tail, dt = parts[0].shape[1:], parts[0].dtype
        for p, d in zip(parts, devices):
            if p.shape[1:] != tail or p.dtype != dt or p.device != d:
                raise ValueError("shards differ in row shape, dtype or device")
        self.parts, self.devices = list(parts), [int(d) for d in devices]
        self.rows = [p.shape[0] for p in parts]
        self.shape, self.dtype = (sum(self.rows),) + tuple(tail), dt

    @staticmethod
    def cut(B, G):
        return [shard_range(B, g, G)[1] - shard_range(B, g, G)[0] for g in range(G)]

    @classmethod
    def empty(cls, shape, dtype, devices, rows=None):
        from ._lib import DeviceArray

        import numpy as np

        rows = cls.cut(shape[0], len(devices)) if rows is None else list(rows)
        return cls([DeviceArray((r,) + tuple(shape[1:]), np.dtype(dtype), d) for r, d in zip(rows, devices)], devices)

    @classmethod
    def zeros(cls, shape, dtype, devices, rows=None):
        out = cls.empty(shape, dtype, devices, rows)
        for p in out.parts:
            if p.shape[0]:
                p.zero_()
        return out

    @classmethod
    def from_numpy(cls, a, devices, dtype=None):
        """scatter a host array once: shard g's rows go to devices[g]"""
        from ._lib import DeviceArray

        import numpy as np

        a = np.ascontiguousarray(a, dtype=dtype)
        G = len(devices)
        parts = []
        for g, d in enumerate(devices):
            lo, hi = shard_range(a.shape[0], g, G)
            parts.append(DeviceArray.from_numpy(a[lo:hi], d))
        return cls(parts, devices)

    def numpy(self, streams=None):
        """gather: the shards' rows in order, as one host array.  streams[g]: the stream shard g's producer ran on (the
        copy is enqueued behind it and waited for); None = the library's own shard streams (abrk_shard_stream), which
        is what the *_resident calls use when no streams are named"""
        import numpy as np

        from ._lib import shard_stream

        out, seen = [], {}
        for g, (p, d) in enumerate(zip(self.parts, self.devices)):
            if streams is None:
                slot = seen.get(d, 0)
                seen[d] = slot + 1
                st = shard_stream(d, slot)
            else:
                st = streams[g]
            out.append(p.numpy(st) if p.shape[0] else np.empty(p.shape, p.dtype))
        return np.concatenate(out)

    def copy_from_numpy(self, a):
        """overwrite the shards with the rows of a host array (a control loop feeding new states into fixed buffers)"""
        import ctypes as C

        import numpy as np

        from ._lib import check, lib

        a = np.ascontiguousarray(a, dtype=self.dtype)
        if a.shape != self.shape:
            raise ValueError(f"expected {self.shape}, got {a.shape}")
        lo = 0
        for p, r in zip(self.parts, self.rows):
            if r:
                h = a[lo:lo + r]
                check(lib().abrk_memcpy_h2d(p.device, p.ptr, h.ctypes.data, h.nbytes, None))
            lo += r
        return self


class ShardedPlan:
    """One recorded control tick per shard (engine.Plan on the shard's device and stream): `launch(K)` / `launch_graph(K)`
    replay K ticks on EVERY device from one call (abrk_plans_launch); `sync()` drains the shards' streams."""

    def __init__(self, plans, streams, like, results):
        self.plans, self.streams, self._like = plans, streams, like
        self.results = results  # whatever the recorded calls returned (ShardedArrays)

    def launch(self, repeat=1):
        from . import engine

        engine.plans_launch(self.plans, repeat, graph=False)

    def launch_graph(self, repeat):
        from . import engine

        engine.plans_launch(self.plans, repeat, graph=True)

    def sync(self):
        from . import engine

        engine.shards_sync(self._like, self.streams)

    def close(self):
        for p in self.plans:
            p.close()
        self.plans = []


class MultiDevice:
    """All (or the named) devices of this process behind one call: `generate(ctrlr, q, dq, target)` cuts the host
    batch into contiguous row shards, evaluates shard g on devices[g] (each on a stream of its own, every kernel in
    flight before the first result is collected) and returns the reassembled NumPy result - BASELINE config 4
    ("2^20 rows sharded across 8 GPUs") as one call.  No collective: rows are independent.  A device may be named
    several times (more shards than devices; how the one-GPU tests exercise the path).

        md = MultiDevice()                  # every visible device
        u = md.generate(ctrlr, Q, dQ, targets)

    `ctrlr`: an abr_control_amd OSC whose secondary controllers are the fused ones (Damping / RestingConfig), or a
    Sliding / Joint / Damping / RestingConfig; `md.dynamics(robot_config, q, ...)` shards the robot_config functions."""

    def __init__(self, devices=None):
        import abr_control_amd as a

        n = a.device_count()
        if n < 1:
            raise RuntimeError("MultiDevice needs a HIP device: abr_control_amd has no CPU fallback")
        self.devices = list(range(n)) if devices is None else [int(d) for d in devices]
        if not self.devices or any(d < 0 or d >= n for d in self.devices):
            raise ValueError(f"devices {self.devices} outside 0..{n - 1}")
        self._streams = None

    # ---- shards that live on the devices
    @property
    def streams(self):
        """the stream of every shard: the library's own (device, slot) streams"""
        if self._streams is None:
            from ._lib import shard_stream

            seen, out = {}, []
            for d in self.devices:
                out.append(shard_stream(d, seen.get(d, 0)))
                seen[d] = seen.get(d, 0) + 1
            self._streams = out
        return self._streams

    def scatter(self, a, dtype=None):
        """host array -> ShardedArray over this object's devices (once; the shards then stay resident)"""
        return ShardedArray.from_numpy(a, self.devices, dtype)

    def empty(self, shape, dtype):
        return ShardedArray.empty(shape, dtype, self.devices)

    def zeros(self, shape, dtype):
        return ShardedArray.zeros(shape, dtype, self.devices)

    def sync(self, like=None):
        """wait for every shard's stream; raises numpy.linalg.LinAlgError once if a shard met a singular inertia matrix"""
        from . import engine

        engine.shards_sync(like if like is not None else _CutOnly(self.devices), self.streams)

    def record(self, fn):
        """Record one control tick per shard: `fn(g, device, stream)` makes the engine calls of shard g (DeviceArrays of
        that device, `stream=stream`) and is run once per shard inside an engine.Plan; or pass a controller through
        `record_generate`.  -> ShardedPlan"""
        from . import engine

        plans, results = [], []
        for g, (d, st) in enumerate(zip(self.devices, self.streams)):
            with engine.Plan(d, st) as plan:
                results.append(fn(g, d, st))
            plans.append(plan)
        return ShardedPlan(plans, self.streams, _CutOnly(self.devices), results)

    def record_generate(self, ctrlr, q, dq, target, target_velocity=None, ref_frame="EE", xyz_offset=None, u=None,
                        training_signal=True):
        """One tick of `ctrlr.generate` on resident ShardedArrays as a ShardedPlan: plan.launch(K) / plan.launch_graph(K)
        replay K ticks on every device from one call; the outputs are plan.u and plan.training_signal (= ctrlr.
        training_signal, as OSC.generate leaves it, osc.py:297; training_signal=False: not computed - the six-row law
        then runs its NOTS kernels, whose u differs from the class's in the last bits, include/abrk.h), per-row state
        (integrated_error) lives in ctrlr.integrated_error as a ShardedArray."""
        from . import engine
        from .controllers import OSC

        if not isinstance(ctrlr, OSC) or not getattr(ctrlr, "_fused_config", False) or ctrlr._foreign or ctrlr._device:
            raise TypeError("record_generate needs an abr_control_amd OSC with fused null controllers only")
        rc = ctrlr.robot_config
        n, B = rc.N_JOINTS, q.shape[0]
        params = ctrlr._params(ref_frame, xyz_offset)
        ie = self._sharded_state(ctrlr, B, rc.dtype)
        u = u if u is not None else ShardedArray.empty((B, n), rc.dtype, self.devices, rows=q.rows)
        ts = training_signal
        if ts is True:
            ts = ShardedArray.empty((B, n), rc.dtype, self.devices, rows=q.rows)
        elif ts is False:
            ts = None
        if ts is not None:
            ctrlr.training_signal = ts

        def tick(g, d, st):
            engine.osc_generate(rc.arm_id, n, params, q.parts[g], dq.parts[g], target.parts[g],
                                None if target_velocity is None else target_velocity.parts[g],
                                None if ie is None else ie.parts[g], None, u.parts[g],
                                False if ts is None else ts.parts[g], dtype=rc.dtype, device=d, stream=st)

        skip = [g for g, r in enumerate(q.rows) if r == 0]
        if skip:
            raise ValueError("record_generate: every shard needs at least one row")
        plan = self.record(tick)
        plan.u, plan.training_signal = u, ts
        plan._keep = (q, dq, target, target_velocity, ie, params)
        return plan

    def _sharded_state(self, ctrlr, B, dtype):
        if ctrlr.ki == 0:
            return None
        ie = ctrlr.integrated_error
        if not isinstance(ie, ShardedArray) or ie.shape != (B, 6) or ie.devices != self.devices:
            ie = ctrlr.integrated_error = ShardedArray.zeros((B, 6), dtype, self.devices)
        return ie

    def generate(self, ctrlr, q, dq, *args, **kwargs):
        """ctrlr.generate(q, dq, ...) with the host batch cut over this object's devices.  OSC (fused null controllers),
        Sliding, Joint, Damping, RestingConfig; arguments and results as the controller's own generate()."""
        from .controllers import OSC

        if isinstance(q, ShardedArray):
            return self._generate_resident(ctrlr, q, dq, *args, **kwargs)
        if isinstance(ctrlr, OSC):
            return self._generate_osc(ctrlr, q, dq, *args, **kwargs)
        from .controllers import Damping, Joint, RestingConfig, Sliding

        # (every Controller inherits _joint_generate: the test is the class, not the attribute)
        if not isinstance(ctrlr, (Sliding, Joint, Damping, RestingConfig)):
            raise TypeError(f"MultiDevice.generate: {type(ctrlr).__name__} has no sharded entry point")
        from ._lib import DeviceArray

        if isinstance(q, DeviceArray):
            raise TypeError("MultiDevice takes NumPy arrays (a DeviceArray lives on one device)")
        ctrlr._shard_devices = self.devices
        try:
            return ctrlr.generate(q, dq, *args, **kwargs)
        finally:
            ctrlr._shard_devices = None

    def dynamics(self, robot_config, q, dq=None, name="EE", x=None, want=("Tx", "J", "M", "g")):
        """robot_config.dynamics(...) - several robot_config functions from one launch per device - over this
        object's devices: {name: [B, ...]} in the kernel dtype"""
        import numpy as np

        from . import engine

        rc = robot_config
        q2, dq2, single, dev = rc._prep(q, dq)
        if dev:
            raise TypeError("MultiDevice takes NumPy arrays (a DeviceArray lives on one device)")
        xo = None if x is None or np.allclose(x, 0) else np.asarray(x, dtype=float)
        res = engine.dynamics_sharded(rc.arm_id, rc.N_JOINTS, np.ascontiguousarray(q2), self.devices,
                                      None if dq2 is None else np.ascontiguousarray(dq2), rc.frame_id(name), xo,
                                      tuple(want), rc.dtype)
        return {k: v[0] for k, v in res.items()} if single else res

    def _generate_resident(self, ctrlr, q, dq, target=None, target_velocity=None, *args, **kwargs):
        """ShardedArray in, ShardedArray out: the call only enqueues (md.sync() / .numpy() wait)"""
        from . import engine
        from .controllers import OSC, Damping, Joint, RestingConfig, Sliding

        rc = ctrlr.robot_config
        n, B = rc.N_JOINTS, q.shape[0]
        if q.devices != self.devices:
            raise ValueError("the ShardedArray is cut over other devices than this MultiDevice")
        if isinstance(ctrlr, OSC):
            if not getattr(ctrlr, "_fused_config", False) or ctrlr._foreign or ctrlr._device:
                raise TypeError("MultiDevice.generate needs an abr_control_amd OSC with fused null controllers only")
            ref_frame = kwargs.pop("ref_frame", args[0] if args else "EE")
            xyz_offset = kwargs.pop("xyz_offset", args[1] if len(args) > 1 else None)
            ie = self._sharded_state(ctrlr, B, rc.dtype)
            u, ts = engine.osc_generate_resident(rc.arm_id, n, ctrlr._params(ref_frame, xyz_offset), q, dq, target,
                                                 target_velocity, ie, None, kwargs.pop("u", None), True, rc.dtype,
                                                 self.streams)
            ctrlr.training_signal = ts
            return u
        from . import _abi

        if isinstance(ctrlr, Sliding):
            names = ("target_acc", "ref_frame", "offset")
            kw = dict(zip(names, args))
            kw.update(kwargs)
            zero = lambda v: None if (v is None or (not isinstance(v, ShardedArray) and v == 0)) else v
            params = _abi.make_sliding_params(n, ctrlr.kd, ctrlr.lamb, ctrlr.cartesian, kw.get("ref_frame", "EE"),
                                              kw.get("offset"))
            u, s = engine.sliding_generate_resident(rc.arm_id, n, params, q, dq, target, zero(target_velocity),
                                                    zero(kw.get("target_acc")), None, True, rc.dtype, self.streams)
            ctrlr.s = s
            return u
        if isinstance(ctrlr, RestingConfig):
            return engine.joint_generate_resident(rc.arm_id, n, ctrlr._ctrl(), False, q, dq, None, None, None, rc.dtype,
                                                  self.streams)
        if isinstance(ctrlr, Damping):
            return engine.joint_generate_resident(rc.arm_id, n, _abi.make_damping(ctrlr.kv), False, q, dq, None, None,
                                                  None, rc.dtype, self.streams)
        if isinstance(ctrlr, Joint):
            return engine.joint_generate_resident(rc.arm_id, n, ctrlr._ctrl(), ctrlr.account_for_gravity, q, dq, target,
                                                  target_velocity, None, rc.dtype, self.streams)
        raise TypeError(f"MultiDevice.generate: {type(ctrlr).__name__} has no resident entry point")

    def _generate_osc(self, ctrlr, q, dq, target, target_velocity=None, ref_frame="EE", xyz_offset=None):
        import numpy as np

        from . import engine

        rc = ctrlr.robot_config
        if not getattr(ctrlr, "_fused_config", False) or ctrlr._foreign or ctrlr._device:
            raise TypeError("MultiDevice.generate needs an abr_control_amd OSC with fused null controllers only")
        (q2, dq2, t2, tv2), single = ctrlr._rows(q, dq, target, target_velocity)
        B = q2.shape[0]
        ie = None
        if ctrlr.ki != 0:
            ie = ctrlr._host_integrated_error(B, rc.dtype)  # a single state's (6,) is kept, as in OSC.generate
        u, ts = engine.osc_generate_sharded(rc.arm_id, rc.N_JOINTS, ctrlr._params(ref_frame, xyz_offset), q2, dq2, t2,
                                            self.devices, tv2, ie, training_signal=True, dtype=rc.dtype)
        if ctrlr.ki != 0:
            ctrlr.integrated_error = ie[0] if single else ie
        if rc.reference_dtypes:
            u, ts = u.astype(np.float64), ts.astype(np.float64)
        ctrlr.training_signal = ts[0] if single else ts
        return u[0] if single else u


class _CutOnly:
    """the shape of a cut without data: what shards_sync needs (devices; one row per shard so that none is skipped)"""

    def __init__(self, devices):
        import numpy as np

        self.devices, self.rows, self.dtype = list(devices), [1] * len(devices), np.dtype(np.float64)
