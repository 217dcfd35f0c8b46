"""Synthetic user arms (1..7 joints) for the runtime-table kernels: random static transforms - exactly
orthogonal or slightly non-orthogonal like Jaco2's truncated constants - random inertia diagonals, an EE
offset, and an N_LINKS that may stop short of the last link (like the reference's onejoint/jaco2 quirks)."""
import numpy as np


def _rot(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def make_arm(n, seed, non_orthogonal=False, n_links_dyn=None):
    rng = np.random.RandomState(seed)

    def aff():
        R = _rot(rng)
        if non_orthogonal:
            R = np.round(R, 6)  # truncated decimals, as in jaco2/config.py:189-273
        t = rng.uniform(-0.3, 0.3, 3)
        return np.hstack([R, t[:, None]]).tolist()

    md = [[0.0] * 6]
    for l in range(1, n + 1):
        m = rng.uniform(0.3, 4.0)
        md.append([m, m, m] + rng.uniform(0.005, 0.08, 3).tolist())
    return {
        "name": f"synth{n}", "n_joints": n, "n_links_dyn": n + 1 if n_links_dyn is None else n_links_dyn,
        "has_ee": 1, "A0": aff(), "AJ": [aff() for _ in range(n)], "B": [aff() for _ in range(n)], "E": aff(),
        "mdiag": md,
    }
