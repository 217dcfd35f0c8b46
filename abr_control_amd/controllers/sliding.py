"""Sliding-mode controller (abr_control/controllers/sliding.py:6-99)."""
import numpy as np

from .. import _abi, engine
from .._lib import DeviceArray
from .controller import Controller


class Sliding(Controller):
    def __init__(self, robot_config, kd=160.0, lamb=30.0, cartesian=True):
        super().__init__(robot_config)
        self._require_batched_config()
        self.kd = kd
        self.lamb = lamb
        self.cartesian = cartesian
        self.s = None

    def generate(self, q, dq, target, target_velocity=0, target_acc=0, ref_frame="EE", offset=None):
        rc = self.robot_config
        n = rc.N_JOINTS
        nt = 3 if self.cartesian else n
        params = _abi.make_sliding_params(n, self.kd, self.lamb, self.cartesian, ref_frame, offset)

        def bc(v):  # the reference accepts scalars (default 0) or arrays (sliding.py:38-39)
            if isinstance(v, DeviceArray):
                return v
            if np.ndim(v) == 0:
                return None if v == 0 else np.full(nt, float(v))
            return v

        (q2, dq2, t2, tv2, ta2), single = self._rows(q, dq, target, bc(target_velocity), bc(target_acc))
        if self._shard_devices:  # under sharding.MultiDevice: contiguous row shards over several devices
            u, s = engine.sliding_generate_sharded(rc.arm_id, n, params, q2, dq2, t2, self._shard_devices, tv2, ta2,
                                                   want_s=True, dtype=rc.dtype)
        else:
            u, s = engine.sliding_generate(rc.arm_id, n, params, q2, dq2, t2, tv2, ta2, want_s=True, dtype=rc.dtype,
                                           device=rc.device)
        if isinstance(u, DeviceArray):
            self.s = s
            return u
        if rc.reference_dtypes:
            u, s = u.astype(np.float64), s.astype(np.float64)
        self.s = s[0] if single else s
        return u[0] if single else u
