#!/usr/bin/env python3
"""A few launches each of the kernels whose executed-instruction counts a floor argument needs (run under
`rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --kernel-trace`: tools/gpu_r6_pmc_legs.sh): for one arm at 1 M rows -
the robot_config functions alone (Tx, J, M, g: kinematics + the inertia matrix), with the velocity-dependent ones (C, dJ:
the full Christoffel matrix), Sliding (which needs all of them + pinv(J[:3]) + the law), the x,y,z OSC law without / with
the Coriolis VECTOR, AvoidObstacles.  usage: pmc_legs.py <arm> [rows]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import abr_control_amd as a  # noqa: E402
from abr_control_amd import _abi, engine  # noqa: E402
from abr_control_amd._lib import check, lib  # noqa: E402

arm = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
n = _abi.load_table(arm)["n_joints"]
arm_id = check(lib().abrk_arm_builtin(arm.encode()))
rng = np.random.RandomState(1)
q, dq, t = rng.uniform(0, 2 * np.pi, (B, n)), rng.uniform(0, 5, (B, n)), rng.uniform(-1, 1, (B, 6))
st = a.Stream(0)
qd, dqd, td = (a.DeviceArray.from_numpy(x) for x in (q, dq, t))
t3 = a.DeviceArray.from_numpy(np.ascontiguousarray(t[:, :3]))
u = a.DeviceArray((B, n))


def rep(fn):
    for _ in range(3):
        fn()
    st.sync()


outs = {w: a.DeviceArray((B,) + s) for w, s in (("Tx", (3,)), ("J", (6, n)), ("M", (n, n)), ("g", (n,)), ("C", (n, n)), ("dJ", (6, n)))}
rep(lambda: engine.dynamics(arm_id, n, qd, None, None, None, ("Tx", "J", "M", "g"), np.float64, 0, st, out=outs))
rep(lambda: engine.dynamics(arm_id, n, qd, dqd, None, None, ("Tx", "J", "M", "g", "C", "dJ"), np.float64, 0, st, out=outs))
rep(lambda: engine.sliding_generate(arm_id, n, _abi.make_sliding_params(n), qd, dqd, t3, u=u, device=0, stream=st))
rep(lambda: engine.osc_generate(arm_id, n, _abi.make_osc_params(n, kp=200), qd, dqd, td, u=u, device=0, stream=st))
rep(lambda: engine.osc_generate(arm_id, n, _abi.make_osc_params(n, kp=200, use_C=True), qd, dqd, td, u=u, device=0, stream=st))
if n >= 3:
    P = _abi.make_obstacles_params(obstacles=[[0.3, 0.2, 0.4, 0.1], [-0.2, 0.4, 0.3, 0.05], [0.1, -0.3, 0.6, 0.15]], threshold=0.3, gain=30)
    rep(lambda: engine.avoid_obstacles_generate(arm_id, n, P, qd, u=u, device=0, stream=st))
rep(lambda: engine.joint_generate(arm_id, n, _abi.make_joint(50, 7), True, qd, dqd, qd, None, u=u, device=0, stream=st))
print("ok", arm, B)
