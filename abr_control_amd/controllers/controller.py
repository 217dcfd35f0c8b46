import numpy as np


class Controller:
    """Base of all controllers (abr_control/controllers/controller.py:4-32)."""

    # set by sharding.MultiDevice for the duration of one generate(): the devices a host batch is cut over
    _shard_devices = None

    def __init__(self, robot_config):
        self.robot_config = robot_config
        self.offset_zeros = np.zeros(3)

    def generate(self, q, dq):
        raise NotImplementedError

    # ---- shared plumbing for the batched device calls
    def _rows(self, *arrays):
        """Normalise (n,) / (B,n) NumPy inputs to 2-D; DeviceArrays pass through.
        Returns (arrays, single)"""
        from .._lib import DeviceArray

        if isinstance(arrays[0], DeviceArray):
            return arrays, False
        dt = self.robot_config.dtype
        single = np.ndim(arrays[0]) == 1
        out = []
        B = np.atleast_2d(np.asarray(arrays[0])).shape[0]
        for a in arrays:
            if a is None:
                out.append(None)
                continue
            a = np.atleast_2d(np.asarray(a, dtype=dt))
            if a.shape[0] == 1 and B > 1:  # one target for the whole batch
                a = np.broadcast_to(a, (B, a.shape[1]))
            out.append(np.ascontiguousarray(a))
        return out, single

    def _joint_generate(self, ctrl, account_for_gravity, q2, dq2, t2=None, tv2=None):
        """the Joint / Damping / RestingConfig kernel on one device, or - under MultiDevice - over several"""
        from .. import engine

        rc = self.robot_config
        if self._shard_devices:
            return engine.joint_generate_sharded(rc.arm_id, rc.N_JOINTS, ctrl, account_for_gravity, q2, dq2,
                                                 self._shard_devices, t2, tv2, dtype=rc.dtype)
        return engine.joint_generate(rc.arm_id, rc.N_JOINTS, ctrl, account_for_gravity, q2, dq2, t2, tv2, dtype=rc.dtype,
                                     device=rc.device)

    def _require_batched_config(self):
        from ..arms.base_config import BatchedConfig

        if not isinstance(self.robot_config, BatchedConfig):
            raise TypeError(
                f"{type(self).__name__} of abr_control_amd evaluates on the GPU and needs an "
                "abr_control_amd arm config (abr_control_amd.arms.<arm>.Config or arms.from_table(...)); "
                f"got {type(self.robot_config).__name__}. There is no CPU fallback."
            )
