"""Joint-space PD controller (abr_control/controllers/joint.py:8-131), angle states."""
import numpy as np

from .. import _abi
from .controller import Controller


class Joint(Controller):
    def __init__(self, robot_config, kp=1, kv=None, quaternions=None, account_for_gravity=True):
        super().__init__(robot_config)
        self._require_batched_config()
        if quaternions is not None:
            raise NotImplementedError(
                "quaternion joint states (joint.py:48-102) only occur with MuJoCo ball joints and are "
                "outside the hot path this package accelerates")
        self.kp = kp
        self.kv = np.sqrt(self.kp) if kv is None else kv
        self.account_for_gravity = account_for_gravity
        self.ZEROS_N_JOINTS = np.zeros(robot_config.N_JOINTS)

    def q_tilde_angle(self, q, target):
        # joint.py:42-46 (host helper; generate() evaluates it inside the kernel)
        return ((target - q + np.pi) % (np.pi * 2)) - np.pi

    def _ctrl(self):
        return _abi.make_joint(self.kp, self.kv)

    def generate(self, q, dq, target, target_velocity=None):
        rc = self.robot_config
        (q2, dq2, t2, tv2), single = self._rows(q, dq, target, target_velocity)
        u = self._joint_generate(self._ctrl(), self.account_for_gravity, q2, dq2, t2, tv2)
        if isinstance(u, np.ndarray) and rc.reference_dtypes:
            u = u.astype(np.float64)
        return u[0] if single else u
