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
        u = self._joint_generate(_abi.make_damping(self.kv), False, q2, dq2)
        if isinstance(u, np.ndarray) and rc.reference_dtypes:
            u = u.astype(np.float64)
        return u[0] if single else u

    def _accumulate(self, q2, dq2, u):
        """u += generate(q2, dq2): used by OSC when more than ABRK_MAX_NULL Damping / RestingConfig controllers are
        given (the first ABRK_MAX_NULL are fused into its kernel; the rest enter through u_null_ext)"""
        if not isinstance(u, np.ndarray):
            raise TypeError(f"more than {_abi.MAX_NULL} fused null controllers need NumPy states")
        rc = self.robot_config
        u += engine.joint_generate(rc.arm_id, rc.N_JOINTS, _abi.make_damping(self.kv), False, q2, dq2, dtype=rc.dtype,
                                   device=rc.device)
