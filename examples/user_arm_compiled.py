"""Bring your own arm (abr_control README: "write a BaseConfig subclass, wait for the code generation"):

1. wherever abr_control is installed, reduce the config to a table of constants - once:
       from tools.extract_arm_table import extract;  json.dump(extract(MyArmConfig()), open("my_arm.json", "w"))
2. here the table runs at once on the runtime-table kernels,
3. and `compiled=True` builds kernels specialised for it (one hipcc run, cached by table values - the counterpart of
   the reference's ~/.cache/abr_control), after which the arm runs at built-in-arm speed.

    python examples/user_arm_compiled.py [my_arm.json]     (needs an MI355X and, for step 3, hipcc)

Without an argument the built-in UR5's table stands in for "my arm", so that the three variants can be compared.
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # run from a checkout

import abr_control_amd as abrk
from abr_control_amd import _abi, arms, engine
from abr_control_amd.controllers import OSC, Damping

table = json.load(open(sys.argv[1])) if len(sys.argv) > 1 else dict(_abi.load_table("ur5"), name="my_arm")
n = table["n_joints"]
rng = np.random.RandomState(1)
B = 1 << 20
Q, DQ, T = rng.uniform(0, 2 * np.pi, (B, n)), rng.uniform(0, 5, (B, n)), rng.uniform(-1, 1, (B, 6))


def rate(rc, label):
    """OSC.generate on device-resident rows, recorded once and replayed"""
    s = abrk.Stream(0)
    dq, ddq, dt_ = (abrk.DeviceArray.from_numpy(a) for a in (Q, DQ, T))
    du = abrk.DeviceArray((B, n))
    with engine.Plan(0, s) as plan:
        engine.osc_generate(rc.arm_id, n, _abi.make_osc_params(n, kp=200), dq, ddq, dt_, u=du, stream=s)
    plan.launch_graph(5)
    s.sync()
    t0 = time.perf_counter()
    plan.launch_graph(20)
    s.sync()
    dt = (time.perf_counter() - t0) / 20
    print(f"{label:28s} {dt * 1e6:8.1f} us per {B} rows = {B / dt / 1e9:6.2f} G control steps/s")
    return du.numpy()


rc_rt = arms.from_table(table, compiled=False)
u_rt = rate(rc_rt, "runtime-table kernels")

t0 = time.perf_counter()
rc = arms.from_table(table, compiled=True)          # builds the plugin on first use, finds it afterwards
rc.arm_id
print(f"compiled kernels ready in {time.perf_counter() - t0:.1f} s: {rc.plugin_path}")
u_c = rate(rc, "compiled kernels")
print("max relative difference of u:", float(np.max(np.abs(u_c - u_rt).max(axis=1) / np.abs(u_rt).max(axis=1))))

# the controller classes take either
ctrlr = OSC(rc, kp=200, null_controllers=[Damping(rc, kv=10)])
print("one state:", ctrlr.generate(Q[0], DQ[0], T[0]))
