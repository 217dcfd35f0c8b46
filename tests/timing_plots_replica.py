#!/usr/bin/env python3
"""The reference's one published benchmark, run on this package and - beside it, on the same machine's host cores - on
the reference itself.

`/root/reference/examples/timing_plots.py:10-39` (the chart `README.rst:159-162` embeds) times `OSC.generate` for one
state at a time: 1000 calls with fresh random `q ~ U(0, 2 pi)`, `dq ~ U(0, 5)`, `target ~ U(-1, 1)^6`, the first call
dropped, the mean reported, per arm: twojoint (OSC defaults), UR5 (`ctrlr_dof = [True] * 6`), Jaco2
(`[True] * 5 + [False]`), Jaco2 with hand.  (The two Jaco2 lines pass `hand_attached=...`, which the reference's
`arms/jaco2/config.py` of today forwards into `BaseConfig.__init__` -> TypeError; the arm as it constructs today stands in
for the first, the second cannot be built - BASELINE.md section 1.)

Three figures per setting:
  * `dropin_ms_per_call`: this package's classes (`abr_control_amd.arms.<arm>.Config`, `abr_control_amd.controllers.OSC`)
    driven by the reference script's own loop - one state per call, NumPy in, NumPy out: a kernel launch, the stream
    synchronisation and the Python / ctypes marshalling around them;
  * `dropin_batched`: the same controller object given [B, n] arrays (what the package is for): ms per call and the
    per-state figure;
  * `reference`: the staged reference (oracle/_ref, oracle/time_reference.py) on one host core of this machine, timed by
    the same per-state loop.
MEASUREMENT TOOLING of the test side (it lives under tests/ because it runs the staged reference through oracle/, which
only tests/, smoke() and bench.py's cpu_baseline leg may touch); run on the GPU box (tools/gpu_timing_plots.sh).  Usage: timing_plots_replica.py [out.json]
"""
import importlib
import json
import os
import sys
import timeit

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

SETTINGS = (  # label, arm module, OSC keyword arguments (timing_plots.py:34-39), staged-reference workload
    ("Two joint", "twojoint", {}, "tp_twojoint"),
    ("UR5", "ur5", {"ctrlr_dof": [True] * 6}, "tp_ur5"),
    ("Jaco2", "jaco2", {"ctrlr_dof": [True] * 5 + [False]}, "tp_jaco2"),
)
N_TRIALS = 1000


def per_state_loop(ctrlr, n_joints):
    """timing_plots.py:14-31, its arithmetic kept: per-call timers, first call dropped, mean of the rest"""
    times = np.zeros(N_TRIALS + 1)
    for ii in range(N_TRIALS + 1):
        q = np.random.random(n_joints) * 2 * np.pi
        dq = np.random.random(n_joints) * 5
        target = np.random.random(6) * 2 - 1
        start = timeit.default_timer()
        ctrlr.generate(q=q, dq=dq, target=target)
        times[ii] = timeit.default_timer() - start
    return float(np.sum(times[1:]) / N_TRIALS)


def batched(ctrlr, n_joints, B, reps=200):
    q = np.random.random((B, n_joints)) * 2 * np.pi
    dq = np.random.random((B, n_joints)) * 5
    target = np.random.random((B, 6)) * 2 - 1
    ctrlr.generate(q=q, dq=dq, target=target)
    t0 = timeit.default_timer()
    for _ in range(reps):
        ctrlr.generate(q=q, dq=dq, target=target)
    return (timeit.default_timer() - t0) / reps


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, "profiles", "round3", "timing_plots.json")
    import abr_control_amd as a
    from abr_control_amd.controllers import OSC

    if a.device_count() < 1:
        sys.exit("timing_plots_replica.py: no HIP device (abr_control_amd has no CPU path)")
    np.random.seed(0)
    res = {"what": "examples/timing_plots.py:10-39 (the reference's published chart, README.rst:159-162): mean wall time "
                   "of one OSC.generate call, one state per call, 1000 calls after the first",
           "published_ms_per_call_cython": {"Two joint": 0.33, "UR5": 0.42, "Jaco2": 0.415},
           "published_note": "read off examples/timing.png (+-5 %), hardware unstated (BASELINE.md section 1)",
           "settings": {}}
    for label, arm, kw, _ in SETTINGS:
        rc = importlib.import_module(f"abr_control_amd.arms.{arm}").Config()
        ctrlr = OSC(rc, **kw)
        per_state_loop(ctrlr, rc.N_JOINTS)  # warm: first pass pages the library in, the second is the measurement
        ms = per_state_loop(ctrlr, rc.N_JOINTS) * 1e3
        row = {"osc_kwargs": {k: [bool(x) for x in v] for k, v in kw.items()}, "dropin_ms_per_call": round(ms, 5),
               "dropin_batched": {}}
        # of which below the Python class (ctypes call -> C ABI -> launch -> stream sync -> back), same kind of states
        from abr_control_amd import engine

        n = rc.N_JOINTS
        qs, dqs, ts = (np.random.random((256, 1, n)) * 2 * np.pi, np.random.random((256, 1, n)) * 5,
                       np.random.random((256, 1, 6)) * 2 - 1)
        prm = ctrlr._params("EE", None)
        t0 = timeit.default_timer()
        for ii in range(N_TRIALS):
            engine.osc_generate(rc.arm_id, n, prm, qs[ii & 255], dqs[ii & 255], ts[ii & 255], None, None, None,
                                training_signal=True, dtype=rc.dtype, device=rc.device)
        row["engine_call_ms"] = round((timeit.default_timer() - t0) / N_TRIALS * 1e3, 5)
        for B in (4096, 65536):
            t = batched(ctrlr, rc.N_JOINTS, B, reps=200 if B == 4096 else 20)
            row["dropin_batched"][str(B)] = {"ms_per_call": round(t * 1e3, 4), "us_per_state": round(t * 1e6 / B, 5)}
        res["settings"][label] = row
        print(label, row, flush=True)
    # the reference itself, same per-state loop, on this machine's host
    from oracle import time_reference

    cores = len(os.sched_getaffinity(0))
    staged = time_reference.measure_staged([s[3] for s in SETTINGS], cores, budget=4.0)
    if staged is None:
        res["reference"] = None
        print("no staged reference here (oracle/_ref missing or sympy/Cython not importable)")
    else:
        res["reference"] = {"measured_on": staged["measured_on"], "script": staged["script"], "settings": {}}
        for label, _, _, wl in SETTINGS:
            w = staged["workloads"][wl]
            res["reference"]["settings"][label] = {"ms_per_call_1core": round(w["us_per_eval_1core"] / 1e3, 5),
                                                    "evals_per_s_allcores": w["evals_per_s_allcores"],
                                                    "function_type": w["function_type"]}
            print("reference", label, res["reference"]["settings"][label], flush=True)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(res, open(out, "w"), indent=1)
    print("wrote", out)


if __name__ == "__main__":
    main()
