"""Iterative inverse-kinematics path generator with the reference's interface
(abr_control/controllers/path_planners/inverse_kinematics.py:7-170), every path's n_timesteps iterations
inside one kernel; `position` / `target_position` may be one path ((n,), (6,)) or a batch ((B,n), (B,6))."""
import numpy as np

from ... import _abi, engine


class InverseKinematics:
    def __init__(self, robot_config, max_dx=0.2, max_dr=2 * np.pi, max_dq=np.pi):
        self.robot_config = robot_config
        self.max_dx = max_dx
        self.max_dr = max_dr
        self.max_dq = max_dq

    def generate_path(self, position, target_position, n_timesteps=200, dt=0.001, plot=False, method=3, axes="rxyz"):
        """Returns (position_path, velocity_path) of shape (n_timesteps, n) - or (B, n_timesteps, n).
        As in the reference the target orientation is read as 'sxyz' Euler angles whatever `axes` says
        (inverse_kinematics.py:73-82 hard-codes axes="sxyz")."""
        if plot:
            raise NotImplementedError("plotting is not part of the accelerated path")
        rc = self.robot_config
        n = rc.N_JOINTS
        single = np.ndim(position) == 1
        pos = np.ascontiguousarray(np.atleast_2d(np.asarray(position, dtype=float)))
        B = pos.shape[0]
        tgt = np.ascontiguousarray(np.broadcast_to(np.atleast_2d(np.asarray(target_position, dtype=float)), (B, 6)))
        p = _abi.make_ik_params(self.max_dx, self.max_dr, self.max_dq, n_timesteps, dt, method)
        pp, vp = engine.ik_generate_path(rc.arm_id, n, p, pos, tgt, device=rc.device)
        self.n_timesteps = n_timesteps
        self.n = 0
        self.position_path = pp[0] if single else pp
        self.velocity_path = vp[0] if single else vp
        return self.position_path, self.velocity_path

    def next(self):
        """next target point along the generated path (inverse_kinematics.py:154-166)"""
        position = self.position_path[..., self.n, :]
        velocity = self.velocity_path[..., self.n, :]
        self.n = min(self.n + 1, self.n_timesteps - 1)
        return position, velocity
