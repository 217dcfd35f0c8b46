"""Batch sharding across GPUs (one process per GPU).  Rows are independent, so a shard is
a contiguous row range and NO collective sits on the data path; what the
ranks of a bench run exchange - a barrier and the maximum of their timings - goes through
HostGroup below: a few bytes over a local socket, no communication library."""
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

    def generate(self, ctrlr, q, dq, *args, **kwargs):
        """ctrlr.generate(q, dq, ...) with the host batch cut over this object's devices.  OSC (fused null controllers),
        Sliding, Joint, Damping, RestingConfig; arguments and results as the controller's own generate()."""
        from .controllers import OSC

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
