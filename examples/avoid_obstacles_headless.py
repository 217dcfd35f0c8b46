"""Headless counterpart of the reference's examples/PyGame/avoid_obstacles.py (lines 12-87) and
force_osc_xy_avoid_joint_limits.py (lines 16-37) on the two-link arm (the reference runs them on its three-link
C++ plant, which is outside this package): OSC in x,y with AvoidObstacles / AvoidJointLimits and Damping behind its
null-space filter, stepped for a whole fleet of arms at once - every generate() below is a handful of kernel launches
for all arms, and the plant step (ArmSim.send_forces) is one more.

    python examples/avoid_obstacles_headless.py        (needs an MI355X)
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # run from a checkout

from abr_control_amd.arms import twojoint as arm  # was: from abr_control.arms import threejoint as arm
from abr_control_amd.controllers import OSC, AvoidJointLimits, AvoidObstacles, Damping

robot_config = arm.Config()
B = 2048
rng = np.random.RandomState(0)
q0 = robot_config.START_ANGLES + rng.uniform(-0.3, 0.3, (B, 2))
fleet = arm.ArmSim(robot_config, q_init=q0)

avoid = AvoidObstacles(robot_config, threshold=1, gain=30)  # avoid_obstacles.py:17
avoid.set_obstacles([[0.8, 1.2, 0, 0.2]])  # one obstacle [x, y, z, radius], shared by all arms
limits = AvoidJointLimits(robot_config, min_joint_angles=[np.pi / 5.0] * 2, max_joint_angles=[np.pi * 0.9] * 2,
                          max_torque=[100.0] * 2)  # force_osc_xy_avoid_joint_limits.py:21-26
damping = Damping(robot_config, kv=10)
ctrlr = OSC(robot_config, kp=10, null_controllers=[avoid, limits, damping], vmax=[10, 0],
            ctrlr_dof=[True, True, False, False, False, False])

targets = np.zeros((B, 6))
targets[:, 0] = rng.uniform(-1.0, 1.0, B)
targets[:, 1] = rng.uniform(1.0, 2.0, B)

t0 = time.perf_counter()
closest = np.full(B, np.inf)
for count in range(1500):
    fb = fleet.get_feedback()
    u = ctrlr.generate(q=fb["q"], dq=fb["dq"], target=targets)  # (B, 2) in -> (B, 2) out
    fleet.send_forces(u)
    if count % 50 == 0:
        hand = robot_config.Tx("EE", fleet.q)
        closest = np.minimum(closest, np.linalg.norm(hand[:, :2] - np.array([0.8, 1.2]), axis=1) - 0.2)
dt = time.perf_counter() - t0
hand = robot_config.Tx("EE", fleet.q)
err = np.linalg.norm(hand[:, :2] - targets[:, :2], axis=1)
print(f"{B} arms x 1500 steps: {dt:.2f} s wall ({B * 1500 / dt / 1e6:.1f} M control steps/s through the Python loop)")
# (the avoidance signal acts in the null space of the x,y task only, as in the reference: a hand whose target lies
#  inside the obstacle still goes there - the links are what is pushed away)
print(f"median distance to target {np.median(err):.3f} m; closest approach of any hand to the obstacle surface "
      f"{closest.min():.3f} m; joint angles within [{fleet.q.min():.2f}, {fleet.q.max():.2f}] rad")
