#!/usr/bin/env python3
"""Prototype (NumPy, CPU) of the early exit of the six-row law's eigen-solver: how many QL rotations does the truncating
pseudo-inverse of Mx_inv really need?

`ql_core` (csrc/abrk_ctrl.h) tridiagonalises the 6 x 6 Mx_inv and runs the implicit QL iteration to ALL six eigenvalues
(~35 rotations).  But a truncating row typically has one or two eigenvalues below rcond * lam_max.  After each deflation
the remaining tridiagonal block T' is tested: if every eigenvalue of T' is certainly kept (all pivots of
LDL^T(T' - cut_hi I) positive, cut_hi from a Gershgorin bound of lam_max) and every eigenvalue found so far is certainly
kept or certainly dropped, the iteration stops and T' is applied as T'^-1 (a tridiagonal solve) - exactly the same
pseudo-inverse, no approximation.  This script counts rotations with and without the exit, with the Wilkinson shift and
with a zero shift on the first pass of each eigenvalue (which steers QL towards the SMALLEST eigenvalue first), on the
Mx_inv of random UR5 states that truncate, and checks the result against numpy.linalg.pinv."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
EPS = np.finfo(float).eps
RCOND = 1e-4


def householder_tridiag(A):
    """as ql_core: H_k acts on rows / columns k+1 .. K-1, k = 0 .. K-3 -> d, e, Q with A = Q T Q^T"""
    K = len(A)
    a = A.copy()
    Q = np.eye(K)
    for k in range(K - 2):
        x = a[k + 1:, k].copy()
        sigma2 = np.sum(x[1:] ** 2)
        if sigma2 > EPS ** 4 * (x[0] ** 2 + sigma2) and sigma2 > 1e-300:
            nrm = np.sqrt(x[0] ** 2 + sigma2)
            alpha = -nrm if x[0] >= 0 else nrm
            v = x.copy()
            v[0] -= alpha
            H = np.eye(K)
            H[k + 1:, k + 1:] -= 2 * np.outer(v, v) / (v @ v)
            a = H @ a @ H
            Q = Q @ H
    return np.diag(a).copy(), np.diag(a, -1).copy(), Q


def exit_test(d, e, l, found):
    """after eigenvalues d[0..l] are final: can the block l+1.. be left as it is?  -> None or ('all_kept'|'all_dropped', cut)"""
    K = len(d)
    blk = range(l + 1, K)
    fmax = max([abs(x) for x in found], default=0.0)
    dmax = max(d[i] for i in blk)
    gersh = max(d[i] + (abs(e[i - 1]) if i - 1 > l else 0.0) + (abs(e[i]) if i < K - 1 else 0.0) for i in blk)
    cut_lo, cut_hi = RCOND * max(fmax, dmax), RCOND * max(fmax, gersh)
    for x in found:  # every found eigenvalue decided whatever lam_max is within its bounds
        if cut_lo < abs(x) <= cut_hi:
            return None
    # all of T' above cut_hi?  Sturm: pivots of LDL^T(T' - cut_hi)
    q = d[l + 1] - cut_hi
    ok = q > EPS * gersh
    for i in range(l + 2, K):
        if not ok:
            break
        q = (d[i] - cut_hi) - e[i - 1] ** 2 / q
        ok = q > EPS * gersh
    return ("all_kept", cut_lo, cut_hi) if ok else None


def ql(d, e, Q, early, zero_first):
    """tql2 on (d, e), rotations accumulated in Q.  -> (n_rotations, l_exit or None)"""
    K = len(d)
    d, e, Q = d.copy(), np.append(e, 0.0), Q.copy()
    rot = 0
    for l in range(K - 1):
        it = 0
        while True:
            m = K - 1
            for j in range(l, K - 1):
                if not abs(e[j]) > EPS * (abs(d[j]) + abs(d[j + 1])):
                    m = j
                    break
            if m == l:
                break
            it += 1
            assert it < 40
            g = (d[l + 1] - d[l]) / (2.0 * e[l])
            r = np.hypot(g, 1.0)
            shift_g = d[m] - d[l] + e[l] / (g + (r if g >= 0 else -r))
            if zero_first and it == 1:
                shift_g = d[m]  # zero shift: g = d[m] - 0
            g = shift_g
            s = c = 1.0
            p = 0.0
            for i in range(m - 1, l - 1, -1):
                f, b = s * e[i], c * e[i]
                r = np.hypot(f, g)
                e[i + 1] = r
                if r == 0.0:
                    d[i + 1] -= p
                    e[m] = 0.0
                    break
                s, c = f / r, g / r
                g = d[i + 1] - p
                r = (d[i] - g) * s + 2.0 * c * b
                p = s * r
                d[i + 1] = g + p
                g = c * r - b
                Qi, Qi1 = Q[:, i].copy(), Q[:, i + 1].copy()
                Q[:, i + 1] = s * Qi + c * Qi1
                Q[:, i] = c * Qi - s * Qi1
                rot += 1
            else:
                d[l] -= p
                e[l] = g
                e[m] = 0.0
        if early and l < K - 2:
            t = exit_test(d, e, l, list(d[:l + 1]))
            if t is not None:
                return rot, l, d, e, Q, t
    return rot, None, d, e, Q, None


def pinv_from(d, e, Q, l_exit, t):
    K = len(d)
    if l_exit is None:
        lam = d
        cut = RCOND * np.max(np.abs(lam))
        w = np.where(np.abs(lam) > cut, 1.0 / np.where(lam == 0, 1, lam), 0.0)
        return (Q * w) @ Q.T
    W = np.zeros((K, K))
    _, cut_lo, cut_hi = t
    for i in range(l_exit + 1):
        W[i, i] = 1.0 / d[i] if abs(d[i]) > cut_hi else 0.0
    nb = K - l_exit - 1
    T = np.diag(d[l_exit + 1:]) + np.diag(e[l_exit + 1:K - 1], 1) + np.diag(e[l_exit + 1:K - 1], -1)
    W[l_exit + 1:, l_exit + 1:] = np.linalg.inv(T) if nb else 0
    return Q @ W @ Q.T


def main():
    from abr_control_amd import _abi
    from oracle.oracle import Oracle

    o = Oracle(_abi.load_table("ur5"))
    rng = np.random.RandomState(3)
    mats = []
    while len(mats) < 400:
        q = rng.uniform(0, 2 * np.pi, 6)
        J, M = o.J("EE", q), o.M(q)
        A = J @ np.linalg.inv(M) @ J.T
        sv = np.linalg.eigvalsh(A)
        if abs(np.linalg.det(A)) < 1e-3 and sv[0] < RCOND * sv[-1] and not np.any(np.abs(sv / sv[-1] - RCOND) < 1e-6 * RCOND):
            mats.append(A)
    print(f"{len(mats)} truncating Mx_inv of random UR5 states")
    for early, zero_first in ((False, False), (True, False), (True, True), (False, True)):
        rots, exits, worst = [], [], 0.0
        for A in mats:
            d, e, Q = householder_tridiag(A)
            rot, l_exit, d2, e2, Q2, t = ql(d, e, Q, early, zero_first)
            P = pinv_from(d2, e2, Q2, l_exit, t)
            ref = np.linalg.pinv(A, rcond=RCOND, hermitian=True)
            worst = max(worst, np.max(np.abs(P - ref)) / np.max(np.abs(ref)))
            rots.append(rot)
            exits.append(-1 if l_exit is None else l_exit)
        rots, exits = np.array(rots), np.array(exits)
        print(f"early={early!s:5} zero_first={zero_first!s:5}: rotations mean {rots.mean():5.1f} median {np.median(rots):4.0f} "
              f"p90 {np.percentile(rots, 90):4.0f} max {rots.max():3d};  exit after eigenvalue #: "
              f"{ {int(k): int((exits == k).sum()) for k in np.unique(exits)} };  worst |pinv - numpy| / |pinv| {worst:.1e}")


if __name__ == "__main__":
    main()
