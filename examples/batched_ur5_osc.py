"""Batched operational-space control of the UR5: the reference API on one state, on a batch, and on
device-resident buffers with a launch plan (the per-tick call of a control loop).

    python examples/batched_ur5_osc.py        (needs an MI355X)
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # run from a checkout

import abr_control_amd as abrk
from abr_control_amd import _abi, engine
from abr_control_amd.arms import ur5
from abr_control_amd.controllers import OSC

rc = ur5.Config()
ctrlr = OSC(rc, kp=200, ctrlr_dof=[True, True, True, False, False, False])
rng = np.random.RandomState(1)

q, dq, target = rng.uniform(0, 2 * np.pi, 6), rng.uniform(0, 5, 6), rng.uniform(-1, 1, 6)
print("one state :", ctrlr.generate(q, dq, target))                 # (6,) float64, as the reference returns

B = 1 << 16
Q, DQ, T = rng.uniform(0, 2 * np.pi, (B, 6)), rng.uniform(0, 5, (B, 6)), rng.uniform(-1, 1, (B, 6))
t0 = time.perf_counter()
U = ctrlr.generate(Q, DQ, T)                                         # (B, 6)
print(f"{B} states through host arrays: {(time.perf_counter() - t0) * 1e3:.2f} ms")

# device-resident: upload once, launch per tick, read back when needed
s = abrk.Stream(0)
dQ, dDQ, dT = (abrk.DeviceArray.from_numpy(a) for a in (Q, DQ, T))
dU = abrk.DeviceArray((B, 6))
plan = engine.OscPlan(rc.arm_id, 6, _abi.make_osc_params(6, kp=200), dQ, dDQ, dT, dU, stream=s)
plan.launch()
s.sync()
t0 = time.perf_counter()
for _ in range(100):
    plan.launch()
s.sync()
dt = (time.perf_counter() - t0) / 100
print(f"{B} states on the device: {dt * 1e6:.1f} us per launch = {B / dt / 1e9:.2f} G control steps/s")
assert np.array_equal(dU.numpy(), U)
