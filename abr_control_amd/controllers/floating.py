"""Gravity compensation (abr_control/controllers/floating.py:6-71), joint space or through the
task-space inertia of the end-effector, optionally cancelling the current momentum."""
import numpy as np

from .. import engine
from .controller import Controller


class Floating(Controller):
    def __init__(self, robot_config, dynamic=False, task_space=False):
        super().__init__(robot_config)
        self._require_batched_config()
        self.dynamic = dynamic
        self.task_space = task_space

    def generate(self, q, dq=None):
        rc = self.robot_config
        if self.dynamic and dq is None:
            raise TypeError("Floating(dynamic=True).generate needs dq")  # floating.py:69 would fail on None
        (q2, dq2), single = self._rows(q, dq if self.dynamic else None)
        u = engine.floating_generate(rc.arm_id, rc.N_JOINTS, self.dynamic, self.task_space, q2, dq2, dtype=rc.dtype,
                                     device=rc.device)
        if isinstance(u, np.ndarray) and rc.reference_dtypes:
            u = u.astype(np.float64)
        return u[0] if single else u

    def _accumulate(self, q2, dq2, u):
        rc = self.robot_config
        engine.floating_generate(rc.arm_id, rc.N_JOINTS, self.dynamic, self.task_space, q2, dq2, u=u,
                                 accumulate=True, dtype=rc.dtype, device=rc.device)
