"""Obstacle avoidance of (Khatib, 1987) (abr_control/controllers/avoid_obstacles.py:6-133): every arm
segment is pushed away from every obstacle closer than `threshold`."""
import numpy as np

from .. import _abi, engine
from .controller import Controller


class AvoidObstacles(Controller):
    def __init__(self, robot_config, obstacles=None, threshold=0.2, gain=1, maximum=500):
        super().__init__(robot_config)
        self._require_batched_config()
        self.threshold = threshold
        self.gain = gain
        self.maximum = maximum
        obstacles = [] if obstacles is None else obstacles
        self.obstacles = np.array(obstacles)

    def set_obstacles(self, obstacles):
        """[[x, y, z, radius], ...] shared by every row of the batch (avoid_obstacles.py:122-133)"""
        self.obstacles = np.copy(obstacles)

    def _params(self):
        return _abi.make_obstacles_params(self.obstacles, self.threshold, self.gain, self.maximum)

    def generate(self, q, dq=None):
        rc = self.robot_config
        (q2,), single = self._rows(q)
        u = engine.avoid_obstacles_generate(rc.arm_id, rc.N_JOINTS, self._params(), q2, dtype=rc.dtype,
                                            device=rc.device)
        if isinstance(u, np.ndarray) and rc.reference_dtypes:
            u = u.astype(np.float64)
        return u[0] if single else u

    def _accumulate(self, q2, dq2, u):
        rc = self.robot_config
        engine.avoid_obstacles_generate(rc.arm_id, rc.N_JOINTS, self._params(), q2, u=u, accumulate=True,
                                        dtype=rc.dtype, device=rc.device)
