"""Headless counterpart of the reference's examples/PyGame/force_osc_xy.py (lines 14-78): the same
controller set-up and control loop - only the imports change (abr_control -> abr_control_amd) and the
PyGame display is dropped.  Runs ONE arm step by step (the reference's loop shape), then 4096 arms for
one simulated second inside a single kernel launch.

    python examples/force_osc_xy_headless.py        (needs an MI355X)
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # run from a checkout

from abr_control_amd.arms import twojoint as arm  # was: from abr_control.arms import twojoint as arm
from abr_control_amd.controllers import OSC, Damping, RestingConfig  # was: abr_control.controllers

robot_config = arm.Config()
arm_sim = arm.ArmSim(robot_config)

damping = Damping(robot_config, kv=10)
resting_config = RestingConfig(robot_config, kp=50, kv=np.sqrt(50), rest_angles=[np.pi / 4, np.pi])
ctrlr = OSC(robot_config, kp=20, use_C=True, null_controllers=[damping, resting_config],
            ctrlr_dof=[True, True, False, False, False, False])

arm_sim.connect()
feedback = arm_sim.get_feedback()
target_xyz = robot_config.Tx("EE", feedback["q"]) + np.array([-0.6, 0.4, 0.0])

# ---- the reference's loop, one arm, one call per millisecond of simulated time
t0 = time.perf_counter()
for count in range(500):
    feedback = arm_sim.get_feedback()
    hand_xyz = robot_config.Tx("EE", feedback["q"])
    target = np.hstack([target_xyz, np.zeros(3)])
    u = ctrlr.generate(q=feedback["q"], dq=feedback["dq"], target=target)
    arm_sim.send_forces(u)
dt = time.perf_counter() - t0
err = np.linalg.norm(robot_config.Tx("EE", arm_sim.q)[:2] - target_xyz[:2])
print(f"1 arm, 500 steps, step by step: {dt * 1e3:.1f} ms wall, distance to target {err:.4f} m")

# ---- the same loop for 4096 arms with different targets, 1000 steps in one launch
B = 4096
rng = np.random.RandomState(0)
q0 = robot_config.START_ANGLES + rng.uniform(-0.5, 0.5, (B, 2))
targets = np.zeros((B, 6))
targets[:, :2] = rng.uniform(-1.5, 1.5, (B, 2))
fleet = arm.ArmSim(robot_config, q_init=q0)
t0 = time.perf_counter()
fleet.rollout(ctrlr, targets, n_steps=1000)
dt = time.perf_counter() - t0
err = np.linalg.norm(robot_config.Tx("EE", fleet.q)[:, :2] - targets[:, :2], axis=1)
print(f"{B} arms, 1000 steps, one launch: {dt * 1e3:.1f} ms wall ({B * 1000 / dt / 1e6:.0f} M control steps/s), "
      f"median distance to target {np.median(err):.4f} m")
