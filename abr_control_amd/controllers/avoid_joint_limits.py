"""Joint-limit avoidance (abr_control/controllers/avoid_joint_limits.py:6-142): a wall or an
exponential gradient of torque pushing each joint back inside [min, max]."""
import numpy as np

from .. import _abi, engine
from .controller import Controller


class AvoidJointLimits(Controller):
    def __init__(self, robot_config, min_joint_angles, max_joint_angles, max_torque=None, cross_zero=None,
                 gradient=None):
        super().__init__(robot_config)
        self._require_batched_config()
        n = robot_config.N_JOINTS
        self._params = _abi.make_limits_params(n, min_joint_angles, max_joint_angles, max_torque, cross_zero,
                                               gradient)
        p = self._params
        # the attributes the reference keeps (avoid_joint_limits.py:53-81)
        nan = np.nan
        self.min_joint_angles = np.array([nan if p.no_limits_min[i] else p.min_joint_angles[i] for i in range(n)])
        self.max_joint_angles = np.array([nan if p.no_limits_max[i] else p.max_joint_angles[i] for i in range(n)])
        self.cross_zero = np.array([bool(p.cross_zero[i]) for i in range(n)])
        self.gradient = np.array([bool(p.gradient[i]) for i in range(n)])
        self.no_limits_min = np.isnan(self.min_joint_angles)
        self.no_limits_max = np.isnan(self.max_joint_angles)
        self.max_torque = np.array([p.max_torque[i] for i in range(n)])

    def generate(self, q, dq=None):
        rc = self.robot_config
        (q2,), single = self._rows(q)
        u = engine.avoid_joint_limits_generate(rc.N_JOINTS, self._params, q2, dtype=rc.dtype, device=rc.device)
        if isinstance(u, np.ndarray) and rc.reference_dtypes:
            u = u.astype(np.float64)
        return u[0] if single else u

    def _accumulate(self, q2, dq2, u):
        """u += generate(q2, dq2) on the device (OSC sums its secondary controllers, osc.py:310-313)"""
        rc = self.robot_config
        engine.avoid_joint_limits_generate(rc.N_JOINTS, self._params, q2, u=u, accumulate=True, dtype=rc.dtype,
                                           device=rc.device)
