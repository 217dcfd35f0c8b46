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
    for fc in cases.fuzz_osc_cases(seed, 24):
        try:
            worst = max(worst, cases.check_fuzz_case(cases.GpuBackend, fc))
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
print(f"soak: {n_sec} secondary-controller cases (seeds {5000 + offset}..{s3 - 1})")
print(f"soak: {n_osc} OSC cases (seeds {100 + offset}..{seed - 1}), worst rel err {worst:.3e}; {n_other} other cases "
      f"(seeds {1000 + offset}..{s2 - 1}); {len(fails)} failures in {time.time() - t0:.0f} s")
for f in fails[:20]:
    print("FAIL", f)
sys.exit(1 if fails else 0)
