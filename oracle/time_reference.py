#!/usr/bin/env python3
"""Times the REFERENCE's own path (abr_control's Cython-loaded generated functions + its NumPy control law, exactly
what its users run) on the host cores of the machine this script is run on, for the BASELINE workloads.

TEST / MEASUREMENT INFRASTRUCTURE.  Two ways to find the reference:
  * build container: /root/reference (scratch copy) + its function cache in ~/.cache/abr_control;
  * anywhere else (the GPU box has no reference checkout): the archive oracle/_ref/abr_control_ref.tar.gz that
    oracle/stage_reference.py packed in the build container - the reference's package and the shared objects its own
    code generator produced.  bench.py's `cpu_baseline` leg calls measure_staged() for that.

The reference has no batch API and no multi-core path: "1 core" is its loop over the batch,
`ctrlr.generate(q[b], dq[b], target[b])` per row; "all cores" is one reference process per core (one arm per
process - its function cache collides across arms, base_config.py:178-191), the sample split evenly.  Inputs: the
reference benchmark's distribution (examples/timing_plots.py:18-20), seed 1 - the same rows bench.py's synthetic
inputs start with.

Usage: python oracle/time_reference.py [out.json] [--staged]    (default out: profiles/round3/reference_cython_baseline.json)
"""
import json
import os
import platform
import shutil
import subprocess
import sys
import tarfile
import tempfile
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
SCRATCH = "/tmp/abrk_ref_scratch_timing"

WORKLOADS = {
    # name: (arm, controller factory source, n target columns)
    "cfg1": ("twojoint", "OSC(rc, kp=10, kv=3, ctrlr_dof=[True, True, False, False, False, False])", 6),
    "cfg2": ("ur5", "OSC(rc, kp=200, ctrlr_dof=[True, True, True, False, False, False])", 6),
    "cfg3": ("jaco2", "OSC(rc, kp=200, null_controllers=[Damping(rc, kv=10)], ctrlr_dof=[True, True, True, False, False, False])", 6),
    "cfg4": ("ur5", "OSC(rc, kp=200, use_g=True, use_C=True, ctrlr_dof=[True, True, True, False, False, False])", 6),
    "cfg5": ("threejoint", "Sliding(rc)", 3),
    # the reference's one published benchmark (examples/timing_plots.py:34-39; README.rst:159-162): OSC with its default
    # gains, per arm.  Its two Jaco2 lines pass `hand_attached=...`, which today's arms/jaco2/config.py forwards into
    # BaseConfig.__init__ -> TypeError; the arm as it constructs today stands in for the first.
    "tp_twojoint": ("twojoint", "OSC(rc)", 6),
    "tp_ur5": ("ur5", "OSC(rc, ctrlr_dof=[True] * 6)", 6),
    "tp_jaco2": ("jaco2", "OSC(rc, ctrlr_dof=[True] * 5 + [False])", 6),
}

WORKER = r'''
import importlib, sys, time
import numpy as np
arm, factory, nt, n_rows, budget = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5])
mod = importlib.import_module(f"abr_control.arms.{arm}")
from abr_control.controllers import OSC, Damping, Sliding
rc = mod.Config(use_cython=True)
ctrlr = eval(factory)
n = rc.N_JOINTS
rng = np.random.RandomState(1)
q, dq, t = rng.uniform(0, 2 * np.pi, (n_rows, n)), rng.uniform(0, 5, (n_rows, n)), rng.uniform(-1, 1, (n_rows, nt))
ctrlr.generate(q[0], dq[0], t[0])  # first call loads the generated functions (examples/timing_plots.py drops it too)
kind = type(rc._M).__name__
done, t0 = 0, time.perf_counter()
while time.perf_counter() - t0 < budget:
    for b in range(n_rows):
        ctrlr.generate(q[b], dq[b], t[b])
    done += n_rows
print(done, time.perf_counter() - t0, kind)
'''


def numpy2_shim(pkg_root):
    """the ONE edit a scratch copy of the reference gets (the same oracle/gen_golden.py applies): numpy >= 2 rejects
    `numpy.array(..., copy=False)` (abr_control/utils/transformations.py:1225), which breaks robot_config.quaternion and
    with it every OSC with orientation rows -> numpy.asarray.  Returns the number of sites replaced."""
    import re

    path = os.path.join(pkg_root, "abr_control", "utils", "transformations.py")
    src = open(path).read()
    src, n = re.subn(r"numpy\.array\(([^()]*?), dtype=numpy\.float64, copy=False\)",
                     r"numpy.asarray(\1, dtype=numpy.float64)", src)
    open(path, "w").write(src)
    return n


def ref_env(pkg_root, home=None):
    """environment of a reference process: its package on the path, one BLAS thread, HOME = where its function cache
    lives (utils/paths.py:9 expands ~/.cache/abr_control)"""
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", PYTHONPATH=pkg_root, OMP_NUM_THREADS="1",
               OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    if home:
        env["HOME"] = home
    return env


def run_workers(name, procs, budget, env):
    arm, factory, nt = WORKLOADS[name]
    cmd = [sys.executable, "-c", WORKER, arm, factory, str(nt), "256", str(budget)]
    t0 = time.perf_counter()
    ps = [subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for _ in range(procs)]
    outs = [p.communicate() for p in ps]
    wall = time.perf_counter() - t0
    for p, (o, e) in zip(ps, outs):
        if p.returncode:
            raise RuntimeError(f"{name}: reference worker failed:\n{e[-2000:]}")
    rows = [o.strip().splitlines()[-1].split() for o, _ in outs]  # last line: OSC prints a notice for twojoint + xyz
    rate = sum(int(r[0]) / float(r[1]) for r in rows)  # each worker's own timed region (start-up excluded)
    return rate, rows[0][2], wall


def host_description():
    cpu = next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), platform.processor())
    return cpu, len(os.sched_getaffinity(0))


def measure(names, env, cores, budget=4.0):
    res = {}
    for name in names:
        r1, kind, _ = run_workers(name, 1, budget, env)
        rall, _, _ = run_workers(name, cores, budget, env)
        res[name] = {"evals_per_s_1core": round(r1, 1), "us_per_eval_1core": round(1e6 / r1, 2),
                     "evals_per_s_allcores": round(rall, 1), "function_type": kind}
    return res


def measure_staged(names, cores, budget=4.0, archive=None):
    """the staged reference (oracle/_ref/abr_control_ref.tar.gz) timed HERE.  -> dict like the committed JSON, or None
    when there is no archive / the reference's dependencies do not import on this machine"""
    archive = archive or os.path.join(REPO, "oracle", "_ref", "abr_control_ref.tar.gz")
    if not os.path.exists(archive):
        return None
    try:
        import cloudpickle  # noqa: F401 - what abr_control.arms.base_config imports
        import Cython  # noqa: F401
        import sympy  # noqa: F401
    except ImportError:
        return None
    root = tempfile.mkdtemp(prefix="abrk_ref_run_")
    try:
        with tarfile.open(archive) as tf:
            tf.extractall(root)
        env = ref_env(os.path.join(root, "pkg"), os.path.join(root, "home"))
        cpu, affinity = host_description()
        res = {"measured_on": f"this machine's host ({cpu}; {cores} cores granted to the process, affinity {affinity})",
               "what": "abr_control's own Cython path: ctrlr.generate(q[b], dq[b], target[b]) per row (no batch API), "
                       "generated functions loaded from its cache (base_config.py:173-191); the reference travelled as "
                       "oracle/_ref/abr_control_ref.tar.gz (oracle/stage_reference.py)",
               "script": "oracle/time_reference.py", "cores": cores, "workloads": measure(names, env, cores, budget)}
        return res
    finally:
        shutil.rmtree(root, ignore_errors=True)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    out = args[0] if args else os.path.join(REPO, "profiles", "round3", "reference_cython_baseline.json")
    cores = len(os.sched_getaffinity(0))
    if "--staged" in sys.argv or not os.path.isdir(REF):
        res = measure_staged(list(WORKLOADS), cores)
        if res is None:
            sys.exit("time_reference.py: neither /root/reference nor a staged oracle/_ref archive (or sympy/Cython missing)")
    else:
        if os.path.isdir(SCRATCH):
            shutil.rmtree(SCRATCH)
        shutil.copytree(REF, SCRATCH)
        numpy2_shim(SCRATCH)
        cpu, _ = host_description()
        res = {"measured_on": f"build container ({cpu}, {cores} cores) - NOT the GPU box's host",
               "what": "abr_control's own Cython path: ctrlr.generate(q[b], dq[b], target[b]) per row (no batch API), "
                       "generated functions loaded from its cache (base_config.py:173-191)",
               "script": "oracle/time_reference.py", "cores": cores,
               "workloads": measure(list(WORKLOADS), ref_env(SCRATCH), cores)}
        shutil.rmtree(SCRATCH, ignore_errors=True)
    for name, v in res["workloads"].items():
        print(name, v, flush=True)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(res, open(out, "w"), indent=1)
    print("wrote", out)


if __name__ == "__main__":
    main()
