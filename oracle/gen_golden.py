#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE ITSELF.

TEST INFRASTRUCTURE - not part of the product.  Runs only in the build container
(where /root/reference exists); the GPU box only ever sees the committed .npz files.

What it does (SURVEY.md section 7 "Hazards" explains every precaution):
  * copies /root/reference to a scratch dir (imports write __pycache__ / pyximport .so),
    sets PYTHONDONTWRITEBYTECODE=1 and PYTHONPATH to the scratch copy;
  * applies ONE shim to the scratch copy only: numpy>=2 rejects
    `numpy.array(..., copy=False)` (abr_control/utils/transformations.py:1225), which
    breaks robot_config.quaternion -> replaced by numpy.asarray;
  * runs ONE ARM PER PROCESS (the reference's function cache collides across arms,
    abr_control/arms/base_config.py:178-191);
  * for every arm stores the raw fp64 outputs of the reference's generated functions
    (Tx, J, M, g, C, dJ, R, T for every frame it offers) on seeded random states;
  * for every controller case stores TWO oracles:
      Oracle-S "as shipped": public ctrlr.generate(...) through the float32-casting
                wrappers (base_config.py:223,247,270,285,301,336),
      Oracle-D "fp64 formulas": the SAME unmodified controller classes driven by a
                robot_config whose wrappers skip the float32 cast;
    plus, per state, det / singular values of Mx_inv so tests can treat states that sit
    within float noise of the two `_Mx` thresholds (osc.py:138,145) separately;
  * stores the closed-form twojoint known answers of the reference's own test fixture
    (abr_control/arms/tests/dummy_base_arm.py) on grids like test_base_config.py:40-180.

Usage:  python oracle/gen_golden.py [--out DIR] [what ...]    (default: tests/golden, all arms + known answers)
        what = <arm> | known | sec:<arm> (secondary controllers, sec_<arm>.npz) |
               quat:<arm> (R and quaternion of every frame, quat_<arm>.npz) |
               helpers:<arm> (OSC._Mx / _velocity_limiting / _calc_orientation_forces, oschelpers_<arm>.npz)
"""
import os
import re
import shutil
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
SCRATCH = "/tmp/abrk_ref_scratch"
OUT = os.path.join(REPO, "tests", "golden")
ARMS = ["twojoint", "threejoint", "ur5", "jaco2", "onejoint"]


def make_scratch():
    if not os.path.isdir(REF):
        sys.exit("gen_golden.py needs /root/reference (build container only)")
    if os.path.isdir(SCRATCH):
        shutil.rmtree(SCRATCH)
    shutil.copytree(REF, SCRATCH)
    path = os.path.join(SCRATCH, "abr_control", "utils", "transformations.py")
    src = open(path).read()
    src, n = re.subn(
        r"numpy\.array\(([^()]*?), dtype=numpy\.float64, copy=False\)",
        r"numpy.asarray(\1, dtype=numpy.float64)",
        src,
    )
    open(path, "w").write(src)
    print(f"scratch copy at {SCRATCH}; numpy-2 shim applied at {n} sites")


def main():
    global OUT
    args = sys.argv[1:]
    if args[:1] == ["--out"]:  # tests/test_reference_provenance.py regenerates into a scratch directory and diffs
        OUT, args = os.path.abspath(args[1]), args[2:]
    which = args or ARMS + ["known"]
    make_scratch()
    env = dict(os.environ)
    env["PYTHONDONTWRITEBYTECODE"] = "1"
    env["PYTHONPATH"] = SCRATCH
    os.makedirs(OUT, exist_ok=True)
    for w in which:
        print(f"==== {w}", flush=True)
        subprocess.run(
            [sys.executable, os.path.join(REPO, "oracle", "_gen_golden_worker.py"), w, OUT],
            env=env,
            check=True,
        )


if __name__ == "__main__":
    main()
