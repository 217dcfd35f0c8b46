#!/usr/bin/env python3
"""Derive an abrk arm table (the constants of include/abrk.h `abrk_arm_desc`) from any
abr_control-style `BaseConfig` subclass by evaluating its `_calc_T(name)` SymPy matrices.

This is the "bring your own arm config" path (reference README.rst:78-87):
instead of SymPy code generation + Cython compilation on first use
(abr_control/arms/base_config.py:125-146), the arm's frame chain is reduced once to a
small table of static transforms that the HIP kernels consume.

Assumed chain structure (true for every arm the reference ships, e.g.
abr_control/arms/ur5/config.py:301-339):
    T(link0)    = A0
    T(joint_i)  = T(link_i) * AJ[i]                   (static)
    T(link_i+1) = T(joint_i) * Rz(q_i) * B[i]         (static after the joint rotation)
    T(EE)       = T(link_n) * E                        (static, may be identity)
The tool verifies the assumption numerically at several random q and refuses otherwise.

Usage (build container, reference importable):
    PYTHONPATH=/path/to/abr_control_checkout python tools/extract_arm_table.py ur5 out.json
or from Python:  table = extract(robot_config)
"""
import json
import re
import sys

import numpy as np


def _snap(v):
    """Recover the literal the config author typed from a value carrying ~1e-16 of
    matrix-inverse noise: either a short decimal, or (threejoint's float32 `L`,
    abr_control/arms/threejoint/config.py:52-67) an exact float32 value."""
    a = float(np.round(v, 10)) + 0.0  # the configs' literals carry <= 10 decimals
    if abs(a - v) < 1e-13:
        return a
    b = float(np.float32(v))
    if abs(b - v) < 1e-13 and b != 0.0:
        return b
    # neither: a constant that is not a short literal (e.g. cos(pi/4) in a non-axis-aligned static rotation).  The
    # runtime tables take arbitrary doubles, so keep the value as evaluated - it then carries the ~1e-16 noise of the
    # matrix solve that isolated it, which is far below the 1e-6 parity bound.
    import warnings

    warnings.warn(f"extract_arm_table: keeping {v!r} as evaluated (not a <=10-digit decimal or a float32 value)")
    return float(v)


def _snapm(M):
    return [[_snap(M[r, c]) for c in range(4)] for r in range(3)]


def _rz(q):
    c, s = np.cos(q), np.sin(q)
    return np.array([[c, -s, 0, 0], [s, c, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1.0]])


def extract(rc, name=None):
    import sympy as sp

    n = int(rc.N_JOINTS)
    names = []
    for i in range(n + 1):
        names.append(f"link{i}")
        if i < n:
            names.append(f"joint{i}")
    names.append("EE")
    fns = {nm: sp.lambdify(rc.q, rc._calc_T(nm), "numpy") for nm in names}

    def T(nm, q):
        return np.array(fns[nm](*q), dtype=float).reshape(4, 4)

    rng = np.random.RandomState(12345)
    tables = []
    for trial in range(3):
        q = rng.uniform(-3, 3, n)
        A0 = T("link0", q)
        AJ, B = [], []
        for i in range(n):
            AJ.append(np.linalg.solve(T(f"link{i}", q), T(f"joint{i}", q)))
            B.append(np.linalg.solve(T(f"joint{i}", q) @ _rz(q[i]), T(f"link{i+1}", q)))
        E = np.linalg.solve(T(f"link{n}", q), T("EE", q))
        tables.append((A0, AJ, B, E))
    for t in tables[1:]:
        for x, y in zip(
            [tables[0][0]] + tables[0][1] + tables[0][2] + [tables[0][3]],
            [t[0]] + t[1] + t[2] + [t[3]],
        ):
            if not np.allclose(x, y, rtol=0, atol=1e-9):
                raise ValueError("frame chain is not of the link/joint/Rz form this tool supports")
    A0, AJ, B, E = tables[0]

    mdiag = []
    for l in range(n + 1):
        if l < len(rc._M_LINKS):
            Ml = np.array(sp.Matrix(rc._M_LINKS[l]).tolist(), dtype=float)
            if not np.allclose(Ml, np.diag(np.diag(Ml))):
                raise ValueError(f"_M_LINKS[{l}] is not diagonal - unsupported")
            mdiag.append([float(v) for v in np.diag(Ml)])
        else:
            mdiag.append([0.0] * 6)
    for Mj in rc._M_JOINTS:
        if np.any(np.array(sp.Matrix(Mj).tolist(), dtype=float) != 0):
            raise ValueError("non-zero _M_JOINTS are not supported")

    has_ee = not np.allclose(E, np.eye(4), rtol=0, atol=1e-12)
    return {
        "name": name or getattr(rc, "ROBOT_NAME", "robot"),
        "n_joints": n,
        "n_links_dyn": int(rc.N_LINKS),
        "has_ee": int(has_ee),
        "A0": _snapm(A0),
        "AJ": [_snapm(m) for m in AJ],
        "B": [_snapm(m) for m in B],
        "E": _snapm(E),
        "mdiag": mdiag,
        "START_ANGLES": [float(v) for v in np.asarray(rc.START_ANGLES, dtype=float)],
    }


if __name__ == "__main__":
    import importlib

    arm, out = sys.argv[1], sys.argv[2]
    mod = importlib.import_module(f"abr_control.arms.{arm}")
    rc = mod.Config(use_cython=False)
    tab = extract(rc, arm)
    txt = json.dumps(tab, indent=1)
    # one matrix row per line
    txt = re.sub(r"\[\s+(-?[0-9.e+-]+),\s+(-?[0-9.e+-]+),\s+(-?[0-9.e+-]+),\s+(-?[0-9.e+-]+)\s+\]",
                 r"[\1, \2, \3, \4]", txt)
    txt = re.sub(r"\[\s+((?:-?[0-9.e+-]+,\s+)+-?[0-9.e+-]+)\s+\]",
                 lambda m: "[" + re.sub(r"\s+", " ", m.group(1)) + "]", txt)
    with open(out, "w") as fh:
        fh.write(txt + "\n")
    print(f"wrote {out}")
