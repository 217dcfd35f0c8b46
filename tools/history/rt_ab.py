#!/usr/bin/env python3
"""A/B of the three ways an arm table reaches the kernels, on the same tables and rows (run on the GPU box): a built-in arm
(compile-time table), a user arm on the runtime-table kernels (abrk_arm_create), and a user arm with its compiled plugin
(abrk_arm_create_compiled, abr_control_amd/specialize.py - build the plugins before the run: they travel in-tree).  Each variant is recorded as a launch plan and replayed as hipGraph launches;
times are HIP-event means per step.  Writes a markdown table (default gpurun_out/r2/rt_ab.md)."""
import ctypes as C
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import abr_control_amd as a  # noqa: E402
from abr_control_amd import _abi, engine, specialize  # noqa: E402
from abr_control_amd._lib import check, lib  # noqa: E402
from abr_control_amd.arms import jaco2, threejoint, ur5  # noqa: E402


def measure(arm, n, p, B, dtype, sliding=False):
    rng = np.random.RandomState(1)
    nt = 3 if sliding else 6
    q, dq, t = rng.uniform(0, 2 * np.pi, (B, n)), rng.uniform(0, 5, (B, n)), rng.uniform(-1, 1, (B, nt))
    s = a.Stream(0)
    qd, dd, td = (a.DeviceArray.from_numpy(x.astype(dtype)) for x in (q, dq, t))
    u = a.DeviceArray((B, n), dtype=dtype)
    with engine.Plan(0, s) as plan:
        if sliding:
            engine.sliding_generate(arm, n, p, qd, dd, td, u=u, stream=s, dtype=dtype)
        else:
            engine.osc_generate(arm, n, p, qd, dd, td, u=u, stream=s, dtype=dtype)
    nodes, reps = (10, 6) if B > 100000 else (100, 40)
    for _ in range(3):
        plan.launch_graph(nodes)
    s.sync()
    e0, e1 = a.Event(0), a.Event(0)
    e0.record(s)
    for _ in range(reps):
        plan.launch_graph(nodes)
    e1.record(s)
    s.sync()
    return e1.elapsed_ms_since(e0) * 1e3 / (reps * nodes), u.numpy()


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, "gpurun_out", "r2", "rt_ab.md")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    lines = ["# Built-in arms vs user arms (runtime table / compiled plugin), same tables, same rows", "",
             f"Device: {a.device_name(0)}.  `tools/rt_ab.py`, hipGraph replay of recorded plans, HIP-event mean per step.", "",
             "| workload | batch | arm | us / step | G steps/s | time vs built-in | max rel diff of u vs built-in |", "|---|---|---|---|---|---|---|"]
    work = [
        ("ur5 osc xyz+g fp64", ur5, "ur5", dict(kp=200), np.float64, False),
        ("ur5 osc xyz+g+C fp64", ur5, "ur5", dict(kp=200, use_C=True), np.float64, False),
        ("ur5 osc 6 rows fp64", ur5, "ur5", dict(kp=200, ctrlr_dof=[1] * 6), np.float64, False),
        ("jaco2 osc xyz+g fp64", jaco2, "jaco2", dict(kp=200), np.float64, False),
        ("threejoint sliding fp32", threejoint, "threejoint", None, np.float32, True),
    ]
    for name, mod, builtin, kw, dtype, sliding in work:
        rc = mod.Config()
        n = rc.N_JOINTS
        static = check(lib().abrk_arm_builtin(builtin.encode()))
        desc = _abi.desc_from_table(rc.table)
        user = check(lib().abrk_arm_create(C.byref(desc)))
        path = specialize.find_compiled(rc.table)
        plug = check(lib().abrk_arm_create_compiled(C.byref(desc), path.encode())) if path else None
        p = _abi.make_sliding_params(n) if sliding else _abi.make_osc_params(n, **kw)
        for B in (4096, 1 << 23):
            us_s, u_s = measure(static, n, p, B, dtype, sliding)
            rows = [("built-in", us_s, u_s), ("user arm, runtime table",) + measure(user, n, p, B, dtype, sliding)]
            if plug is not None:
                rows.append(("user arm, compiled plugin",) + measure(plug, n, p, B, dtype, sliding))
            for tag, us, u in rows:
                diff = np.max(np.max(np.abs(u - u_s), axis=1) / np.max(np.abs(u_s), axis=1))
                lines.append(f"| {name} | {B} | {tag} | {us:.2f} | {B / us / 1e3:.3f} | {us / us_s:.2f}x | {diff:.1e} |")
                print(lines[-1], flush=True)
    open(out, "w").write("\n".join(lines) + "\n")
    print("wrote", out)


if __name__ == "__main__":
    main()
