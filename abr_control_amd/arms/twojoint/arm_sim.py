"""Batched two-link plant with the reference's ArmSim interface
(abr_control/arms/twojoint/arm_sim.py:4-143): connect / disconnect / get_feedback / send_forces / reset,
for one arm (q of shape (2,)) or B arms (q_init of shape (B, 2)), stepped on the GPU; plus `rollout`, which
runs the whole control loop of examples/PyGame/force_osc_xy.py:57-78 - OSC.generate followed by the plant
step - for n_steps in a single kernel launch."""
import numpy as np

from ... import _abi, engine


class ArmSim:
    def __init__(self, robot_config, dt=0.001, q_init=None):
        self.robot_config = robot_config
        self.q_init = np.array(q_init if q_init is not None else robot_config.START_ANGLES, dtype=float)
        self.dt = dt
        self.t = 0.0
        self._plant = _abi.make_twolink_plant(robot_config.L, robot_config._M_LINKS, dt)
        self.K1, self.K2, self.K3, self.K4 = (self._plant.K1, self._plant.K2, self._plant.K3, self._plant.K4)
        self.reset()

    def connect(self):
        self.reset()

    def disconnect(self):
        self.reset()

    def reset(self):
        self.q = np.copy(self.q_init)
        self.dq = np.zeros(self.q.shape)

    def get_feedback(self):
        return {"q": self.q, "dq": self.dq}

    def _state2d(self):
        self._single = self.q.ndim == 1
        q = np.ascontiguousarray(np.atleast_2d(self.q), dtype=float)
        dq = np.ascontiguousarray(np.atleast_2d(self.dq), dtype=float)
        return q, dq

    def _commit(self, q, dq):
        self.q, self.dq = (q[0], dq[0]) if self._single else (q, dq)

    def send_forces(self, u, dt=None):
        """advance one time step under torques u (arm_sim.py:67-82, 101-137)"""
        plant = self._plant if dt is None else _abi.make_twolink_plant(self.robot_config.L,
                                                                        self.robot_config._M_LINKS, dt)
        q, dq = self._state2d()
        engine.twolink_step(plant, q, dq, np.atleast_2d(np.asarray(u, dtype=float)),
                            device=self.robot_config.device)
        self._commit(q, dq)
        self.t += self.dt

    def rollout(self, ctrlr, target, n_steps, every=0):
        """n_steps of { u = ctrlr.generate(q, dq, target); send_forces(u) } in one launch (ctrlr: an
        abr_control_amd OSC on this robot_config).  Returns (q_traj, dq_traj, u_traj) sampled every
        `every` steps, or None."""
        rc = self.robot_config
        q, dq = self._state2d()
        B = q.shape[0]
        t2 = np.ascontiguousarray(np.broadcast_to(np.atleast_2d(np.asarray(target, float)), (B, 6)))
        ie = None
        if ctrlr.ki != 0:
            ie = np.ascontiguousarray(np.broadcast_to(np.atleast_2d(ctrlr.integrated_error), (B, 6)), dtype=float)
        # the fused loop evaluates only what lives inside the OSC kernel: Damping / RestingConfig.  Anything else
        # (AvoidJointLimits, AvoidObstacles, Floating, Python controllers) would be silently dropped - refuse it
        if ctrlr._foreign or ctrlr._device:
            raise TypeError("rollout needs fused null controllers (Damping / RestingConfig); run the per-step "
                            "loop (ctrlr.generate + send_forces) for the others")
        if ctrlr.robot_config is not rc:
            raise ValueError("rollout: the controller was built on a different robot_config than this plant")
        res = engine.osc_rollout_twolink(rc.arm_id, ctrlr._params("EE", None), self._plant, q, dq, t2, n_steps,
                                         every, ie, want_traj=every > 0, device=rc.device)
        self._commit(q, dq)
        if ie is not None:
            ctrlr.integrated_error = ie[0] if self._single else ie
        self.t += self.dt * n_steps
        if res is not None and self._single:
            res = tuple(r[0] for r in res)
        return res
