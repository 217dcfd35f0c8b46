#!/usr/bin/env python3
"""VERDICT r4 "Next" #8: would emulating the reference's float32 rounding points close the gap to its SHIPPED path?

The shipped reference rounds the generated functions' fp64 results to float32 at its wrappers
(abr_control/arms/base_config.py:223 g, :247 dJ, :270 J, :285 M, :301 R, :336 C) and then runs osc.py's NumPy law on
those float32 arrays - `np.linalg.inv(M)` (osc.py:136), `np.dot(J, np.dot(M_inv, J.T))` (:137), `det`, `inv` / `pinv`
(:138-145) are float32 LAPACK / BLAS calls; only what touches the fp64 `Tx` / `q` / `dq` / `target` is promoted.

On the 4096 rows of BASELINE config 2 in tests/golden/ur5.npz (`cfg2_uS` = the reference as shipped, `cfg2_uD` = the same
formulas on its fp64 functions) this script evaluates, with the CPU oracle's fp64 J / M / g / Tx (CPU only, no GPU):

  A  the fp64 law on fp64 J, M, g                  - what the kernels compute (Oracle-D)
  B  the fp64 law on float32-ROUNDED J, M, g       - what an opt-in `reference_rounding` inside the kernels would compute:
                                                     the six rounding points, fp64 arithmetic after them
  C  osc.py's own NumPy expressions on float32 J, M, g - float32 inv / dot / det / pinv exactly as the shipped path calls them

and prints the distance of each from `cfg2_uS` (max|du| / max|u| per row: median, p99, max).  Outcome (committed in
profiles/round5/reference_rounding.md): see there."""
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def law(J, M, g, xyz, q, dq, target, kp=200.0):
    """osc.py:244-301 for ctrlr_dof = x,y,z, kv = sqrt(kp + ko) with ko = kp, use_g, no vmax / ki / use_C / nulls -
    written with the reference's own NumPy calls so that the arrays' dtypes decide the arithmetic, as they do there"""
    kv = np.sqrt(kp + kp)
    J = J[:3]
    M_inv = np.linalg.inv(M)                                # osc.py:136
    Mx_inv = np.dot(J, np.dot(M_inv, J.T))                  # :137
    if abs(np.linalg.det(Mx_inv)) >= 1e-3:                  # :138
        Mx = np.linalg.inv(Mx_inv)
    else:
        Mx = np.linalg.pinv(Mx_inv, rcond=1e-3 * 0.1)       # :145
    u_task = np.zeros(6)
    u_task[:3] = xyz - target[:3]
    u_task *= np.array([kp] * 3 + [kp] * 3)
    u = -1 * kv * np.dot(M, dq)
    u_task = u_task[:3]
    u -= np.dot(J.T, np.dot(Mx, u_task))
    u -= g
    return u


def main():
    from abr_control_amd import _abi
    from oracle.oracle import Oracle

    gold = np.load(os.path.join(REPO, "tests", "golden", "ur5.npz"))
    q, dq, t, uS, uD = (gold[k] for k in ("cfg2_q", "cfg2_dq", "cfg2_target", "cfg2_uS", "cfg2_uD"))
    o = Oracle(_abi.load_table("ur5"))
    B = len(q)
    out = {k: np.empty((B, 6)) for k in "ABC"}
    for b in range(B):
        J, M, g, x = o.J("EE", q[b]), o.M(q[b]), o.g(q[b]), o.Tx("EE", q[b])
        J32, M32, g32 = J.astype(np.float32), M.astype(np.float32), g.astype(np.float32)
        out["A"][b] = law(J, M, g, x, q[b], dq[b], t[b])
        out["B"][b] = law(J32.astype(np.float64), M32.astype(np.float64), g32.astype(np.float64), x, q[b], dq[b], t[b])
        out["C"][b] = law(J32, M32, g32, x, q[b], dq[b], t[b])
    rel = lambda a, ref: np.max(np.abs(a - ref), axis=1) / np.max(np.abs(ref), axis=1)
    res = {"rows": B, "numpy": np.__version__}
    for k, what in (("A", "fp64 law, fp64 J/M/g (the kernels; Oracle-D)"),
                    ("B", "fp64 law, float32-rounded J/M/g (kernel-side reference_rounding)"),
                    ("C", "osc.py's NumPy expressions on float32 J/M/g (float32 inv / dot / det / pinv)")):
        r = rel(out[k], uS)
        res[k] = {"what": what, "median": float(np.median(r)), "p99": float(np.percentile(r, 99)), "max": float(r.max()),
                  "rows_beyond_1e-6": int((r > 1e-6).sum()), "rows_bit_equal": int((out[k] == uS).all(axis=1).sum())}
    res["A_vs_uD_max"] = float(rel(out["A"], uD).max())
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
