#!/usr/bin/env python3
"""Extended seeded fuzz of the GPU path against the oracle (same case generators as tests/test_gpu_parity.py, many
more seeds).  Test tooling: prints every failing case instead of stopping at the first."""
import os
import sys
import time
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import cases  # noqa: E402

t0 = time.time()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
offset = int(sys.argv[2]) if len(sys.argv) > 2 else 0  # shifts every seed range: a different sample per run
fails, n_osc, n_other, worst = [], 0, 0, 0.0
seed = 100 + offset
while time.time() - t0 < budget * 0.6:
    # (each case draws whether it asks for the training signal; every third seed is the plain six-row law without it -
    #  the NOTS kernels bench.py times - in the hand-over and the one-pass form alternately)
    plain = seed % 3 == 2
    factory = (lambda tab, f=("auto", "slices")[(seed // 3) & 1]: cases.GpuBackend(tab, form=f)) if plain else cases.GpuBackend
    for fc in cases.fuzz_osc_cases(seed, 24, plain_six=plain):
        try:
            worst = max(worst, cases.check_fuzz_case(factory, fc))
        except Exception as e:  # noqa: BLE001
            fails.append(("osc", seed, fc.get("kw"), repr(e)[:300]))
        n_osc += 1
    seed += 1
s2 = 1000 + offset
while time.time() - t0 < budget:
    try:
        cases.check_fuzz_other(cases.GpuBackend, s2)
    except Exception as e:  # noqa: BLE001
        fails.append(("other", s2, traceback.format_exc()[-400:]))
    n_other += 1
    s2 += 1
s3, n_sec = 5000 + offset, 0
t3 = time.time()
while time.time() - t3 < budget * 0.4:
    try:
        cases.check_fuzz_secondary(cases.GpuBackend, s3)
    except Exception as e:  # noqa: BLE001
        fails.append(("secondary", s3, repr(e)[:300]))
    n_sec += 1
    s3 += 1
# six-row law, big batches: the deferred second pass (rows whose pinv truncates) against the inline sweeps of chunked
# calls, bit for bit, on random user arms; and the fused u + Tx,J,M,g kernel against the separate calls
import numpy as np  # noqa: E402

from abr_control_amd import _abi  # noqa: E402
from tests.synthetic_arms import make_arm  # noqa: E402

n_def, t4, s4 = 0, time.time(), 9000 + offset
while time.time() - t4 < min(30.0, budget * 0.15):
    rng = np.random.RandomState(s4)
    n = int(rng.randint(4, 8))
    # with or without the training signal asked for (drawn per case): without it the plain six-row law below runs its NOTS
    # instantiations in every pass - what bench.py times
    be = cases.GpuBackend(make_arm(n, s4, non_orthogonal=bool(rng.randint(2))), training_signal=bool(s4 & 1))
    B = 20000
    q, dq, t = rng.uniform(-3, 3, (B, n)), rng.uniform(-2, 2, (B, n)), rng.uniform(-0.6, 0.6, (B, 6))
    dof = [1] * 6 if n >= 6 else [1, 1, 1, 1, 0, 0]
    p = _abi.make_osc_params(n, kp=100, ko=60, kv=12, ctrlr_dof=dof, use_C=bool(rng.randint(2)),
                             orientation_algorithm=int(rng.randint(2)))
    try:
        u_big = be.osc(p, q, dq, t)[0]
        u_c = np.concatenate([be.osc(p, q[lo:lo + 5000], dq[lo:lo + 5000], t[lo:lo + 5000])[0] for lo in range(0, B, 5000)])
        assert np.array_equal(u_big, u_c, equal_nan=True), "deferred pass differs from inline sweeps"
        u1, _, dyn = be.e.osc_generate(be.arm_id, n, p, q[:3000], dq[:3000], t[:3000], training_signal=True,
                                       want=("Tx", "J", "M", "g"))
        ref = be.e.dynamics(be.arm_id, n, q[:3000], None, _abi.frame_id("EE", n), None, ("Tx", "J", "M", "g"), np.float64, 0)
        for k in ref:
            assert np.max(np.abs(dyn[k] - ref[k])) <= 1e-12 * max(1.0, np.max(np.abs(ref[k]))), k
        fin = np.isfinite(u_c[:3000]).all(axis=1) & np.isfinite(u1).all(axis=1)
        assert np.max(np.abs(u1[fin] - u_c[:3000][fin])) <= 1e-9 * max(1.0, np.max(np.abs(u_c[:3000][fin]))), "fused u"
        # round 3: the velocity-dependent outputs from the fused kernel, and the one-call-every-device entry points
        allw = ("Tx", "J", "M", "g", "C", "dJ")
        u2, _, dyn2 = be.e.osc_generate(be.arm_id, n, p, q[:3000], dq[:3000], t[:3000], training_signal=True, want=allw)
        ref2 = be.e.dynamics(be.arm_id, n, q[:3000], dq[:3000], _abi.frame_id("EE", n), None, allw, np.float64, 0)
        for k in allw:
            assert np.max(np.abs(dyn2[k] - ref2[k])) <= 1e-11 * max(1.0, np.max(np.abs(ref2[k]))), "VEL " + k
        fin2 = fin & np.isfinite(u2).all(axis=1)
        assert np.max(np.abs(u2[fin2] - u_c[:3000][fin2])) <= 1e-8 * max(1.0, np.max(np.abs(u_c[:3000][fin2]))), "VEL u"
        devs = [0] * int(rng.randint(1, 6))
        ds = be.e.dynamics_sharded(be.arm_id, n, q[:3000], devs, dq[:3000], _abi.frame_id("EE", n), None, allw)
        assert all(np.array_equal(ds[k], ref2[k], equal_nan=True) for k in allw), "sharded dynamics"
        ps = _abi.make_sliding_params(n)
        us = be.e.sliding_generate(be.arm_id, n, ps, q[:3000], dq[:3000], t[:3000, :3])
        assert np.array_equal(be.e.sliding_generate_sharded(be.arm_id, n, ps, q[:3000], dq[:3000], t[:3000, :3], devs), us,
                              equal_nan=True), "sharded sliding"
    except Exception as e:  # noqa: BLE001
        fails.append(("deferred/fused", s4, n, repr(e)[:300]))
    n_def += 1
    s4 += 1
print(f"soak: {n_def} deferred-pass / fused-output cases (seeds {9000 + offset}..{s4 - 1})")
print(f"soak: {n_sec} secondary-controller cases (seeds {5000 + offset}..{s3 - 1})")
print(f"soak: {n_osc} OSC cases (seeds {100 + offset}..{seed - 1}), worst rel err {worst:.3e}; {n_other} other cases "
      f"(seeds {1000 + offset}..{s2 - 1}); {len(fails)} failures in {time.time() - t0:.0f} s")
for f in fails[:20]:
    print("FAIL", f)
sys.exit(1 if fails else 0)
