"""Move towards resting joint angles in the null space
(abr_control/controllers/resting_config.py:6-42)."""
import numpy as np

from .. import _abi, engine
from .joint import Joint


class RestingConfig(Joint):
    def __init__(self, robot_config, rest_angles, **kwargs):
        super().__init__(robot_config, account_for_gravity=False, **kwargs)
        self.rest_angles_list = list(rest_angles)
        self.rest_angles = np.asarray(rest_angles)
        self.rest_indices = [val is not None for val in rest_angles]

    def q_tilde_angle(self, q, target=None):
        # resting_config.py:25-31
        q = np.asarray(q, dtype=float)
        q_tilde = np.zeros(len(q))
        idx = np.asarray(self.rest_indices)
        rest = np.array([0.0 if v is None else v for v in self.rest_angles_list])
        q_tilde[idx] = (rest[idx] - q[idx] + np.pi) % (np.pi * 2) - np.pi
        return q_tilde

    def _ctrl(self):
        return _abi.make_resting(self.rest_angles_list, self.kp, self.kv)

    def generate(self, q, dq):
        rc = self.robot_config
        (q2, dq2), single = self._rows(q, dq)
        u = self._joint_generate(self._ctrl(), False, q2, dq2)
        if isinstance(u, np.ndarray) and rc.reference_dtypes:
            u = u.astype(np.float64)
        return u[0] if single else u

    def _accumulate(self, q2, dq2, u):
        """u += generate(q2, dq2): used by OSC when more than ABRK_MAX_NULL Damping / RestingConfig controllers are
        given (the first ABRK_MAX_NULL are fused into its kernel; the rest enter through u_null_ext)"""
        if not isinstance(u, np.ndarray):
            raise TypeError(f"more than {_abi.MAX_NULL} fused null controllers need NumPy states")
        rc = self.robot_config
        u += engine.joint_generate(rc.arm_id, rc.N_JOINTS, self._ctrl(), False, q2, dq2, dtype=rc.dtype, device=rc.device)
