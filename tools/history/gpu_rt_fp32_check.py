#!/usr/bin/env python3
"""the fp32 Sliding result of the threejoint arm on the runtime-table kernels differed from the built-in
kernels by up to 2.6e-3 (relative, worst of 8 M rows; profiles/round2/rt_ab.md).  Both are fp32 evaluations of an
ill-conditioned map near the planar arm's singular configurations; this script measures each against the fp64 kernels
on the SAME float32-rounded inputs, row by row: if the two fp32 programs are equally far from fp64, the difference
between them is rounding, not a defect of either."""
import sys

import numpy as np

sys.path.insert(0, ".")
from abr_control_amd import _abi, engine
from abr_control_amd._lib import check, lib
import ctypes as C

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 23
tab = _abi.load_table("threejoint")
p = _abi.make_sliding_params(3)
rng = np.random.RandomState(1)
q, dq, t = (rng.uniform(0, 2 * np.pi, (B, 3)).astype(np.float32), rng.uniform(0, 5, (B, 3)).astype(np.float32),
            rng.uniform(-1, 1, (B, 3)).astype(np.float32))
a_static = check(lib().abrk_arm_builtin(b"threejoint"))
d = _abi.desc_from_table(tab)
a_rt = check(lib().abrk_arm_create(C.byref(d)))
us = engine.sliding_generate(a_static, 3, p, q, dq, t, dtype=np.float32)
ur = engine.sliding_generate(a_rt, 3, p, q, dq, t, dtype=np.float32)
u64 = engine.sliding_generate(a_static, 3, p, q.astype(float), dq.astype(float), t.astype(float), dtype=np.float64)
rel = lambda a, b: np.max(np.abs(a.astype(float) - b), axis=1) / np.max(np.abs(b), axis=1)
es, er, esr = rel(us, u64), rel(ur, u64), rel(us, ur.astype(float))
for name, e in (("built-in fp32 vs fp64", es), ("runtime-table fp32 vs fp64", er), ("built-in vs runtime-table", esr)):
    print(f"{name:28s} median {np.median(e):.2e}  p99 {np.percentile(e, 99):.2e}  p99.99 {np.percentile(e, 99.99):.2e}  max {e.max():.2e}")
good = es <= 1e-5
print(f"rows where the built-in fp32 kernel is within 1e-5 of fp64: {good.mean():.4f}; on them built-in vs runtime-table max "
      f"{esr[good].max():.2e}, runtime-table vs fp64 max {er[good].max():.2e}")
w = np.argsort(esr)[-3:]
print("worst rows (built-in vs runtime-table):", [(int(i), float(esr[i]), float(es[i]), float(er[i]), (q[i] % (2 * np.pi)).tolist()) for i in w])
