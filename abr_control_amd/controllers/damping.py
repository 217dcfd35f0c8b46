"""Null-space damping controller (abr_control/controllers/damping.py:6-32): u = M (-kv dq)."""
import numpy as np

from .. import _abi, engine
from .controller import Controller


class Damping(Controller):
    def __init__(self, robot_config, kv):
        super().__init__(robot_config)
        self._require_batched_config()
        self.kv = kv

    def generate(self, q, dq):
        rc = self.robot_config
        (q2, dq2), single = self._rows(q, dq)
        u = engine.joint_generate(rc.arm_id, rc.N_JOINTS, _abi.make_damping(self.kv), False, q2, dq2,
                                  dtype=rc.dtype, device=rc.device)
        if isinstance(u, np.ndarray) and rc.reference_dtypes:
            u = u.astype(np.float64)
        return u[0] if single else u
