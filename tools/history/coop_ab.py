#!/usr/bin/env python3
"""A/B of the two thread mappings of OSC.generate on the MI355X (run on the GPU box):
lane-per-arm (one lane = one arm instance, abrk_kernels.h) against the wave-cooperative mapping of BASELINE.json's
north_star (K = 4 / 8 / 16 lanes per arm, frames + Jacobian columns + M staged in LDS, abrk_coop.h), for the plain law of
BASELINE config 2 (UR5, fp64, x,y,z of the EE) at the config batch and at larger ones.  Every variant is replayed from a
recorded launch plan as hipGraph launches of 100 kernel nodes (bench.py's protocol); times are HIP-event means per step.
Writes a markdown table to the path given (default gpurun_out/r2/coop_ab.md)."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import abr_control_amd as a  # noqa: E402
from abr_control_amd import _abi, engine  # noqa: E402
from abr_control_amd._lib import check, lib  # noqa: E402


def measure(B, lanes, reps=40, nodes=100):
    arm = check(lib().abrk_arm_builtin(b"ur5"))
    p = _abi.make_osc_params(6, kp=200)
    rng = np.random.RandomState(1)
    q, dq, t = rng.uniform(0, 2 * np.pi, (B, 6)), rng.uniform(0, 5, (B, 6)), rng.uniform(-1, 1, (B, 6))
    s = a.Stream(0)
    qd, dd, td = (a.DeviceArray.from_numpy(x) for x in (q, dq, t))
    u = a.DeviceArray((B, 6))
    with engine.Plan(0, s) as plan:
        if lanes == 1:
            engine.osc_generate(arm, 6, p, qd, dd, td, u=u, stream=s)
        else:
            engine.osc_generate_coop(arm, 6, p, qd, dd, td, lanes, u=u, stream=s)
    if B > 100000:
        nodes, reps = 10, 6
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
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, "gpurun_out", "r2", "coop_ab.md")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    lines = ["# Thread mapping A/B: lane-per-arm vs wave-cooperative (K lanes per arm), UR5 OSC xyz + g, fp64", "",
             f"Device: {a.device_name(0)}.  `tools/coop_ab.py`: each variant recorded as a launch plan and replayed as hipGraph "
             "launches of 100 kernel nodes (10 above 100 k rows); us per step = HIP-event time / steps.  "
             "Wavefronts = what one launch puts on the chip's 1024 SIMDs.", "",
             "| batch | mapping | wavefronts | us / step | G steps/s | vs lane-per-arm | max rel diff of u |", "|---|---|---|---|---|---|---|"]
    for B in (4096, 16384, 65536, 1 << 20):
        base, uref = None, None
        for lanes in (1, 4, 8, 16):
            us, u = measure(B, lanes)
            if lanes == 1:
                base, uref = us, u
            waves = (B + 63) // 64 if lanes == 1 else (B + 64 // lanes - 1) // (64 // lanes)
            diff = np.max(np.max(np.abs(u - uref), axis=1) / np.max(np.abs(uref), axis=1))
            name = "lane-per-arm" if lanes == 1 else f"cooperative, K = {lanes}"
            lines.append(f"| {B} | {name} | {waves} | {us:.2f} | {B / us / 1e3:.3f} | {base / us:.2f}x | {diff:.1e} |")
            print(lines[-1], flush=True)
    open(out, "w").write("\n".join(lines) + "\n")
    print("wrote", out)


if __name__ == "__main__":
    main()
