#!/bin/bash
# host-array (NumPy in / NumPy out) call latency: copy-engine staging vs the pinned zero-copy arena
cd "$GRAFT_REPO_ROOT" || exit 1
cat > /tmp/staged.py <<'PY'
import os, sys, time, numpy as np
from abr_control_amd import _abi, engine
from abr_control_amd._lib import check, lib
arm_id = check(lib().abrk_arm_builtin(b"ur5"))
p = _abi.make_osc_params(6, kp=200)
rng = np.random.RandomState(1)
for B in (1, 64, 1024, 4096, 16384, 65536):
    q, dq, t = rng.uniform(0, 6, (B, 6)), rng.uniform(0, 5, (B, 6)), rng.uniform(-1, 1, (B, 6))
    u = np.empty((B, 6))
    for _ in range(20): engine.osc_generate(arm_id, 6, p, q, dq, t, u=u)
    reps = 2000 if B <= 4096 else 200
    t0 = time.perf_counter()
    for _ in range(reps): engine.osc_generate(arm_id, 6, p, q, dq, t, u=u)
    dt = (time.perf_counter() - t0) / reps
    print(f"  B={B:6d}  {dt*1e6:9.2f} us/call  {B/dt/1e6:9.3f} M evals/s", flush=True)
if len(sys.argv) > 1:
    import cProfile, pstats
    q, dq, t = rng.uniform(0, 6, (1, 6)), rng.uniform(0, 5, (1, 6)), rng.uniform(-1, 1, (1, 6))
    from abr_control_amd.arms import ur5
    from abr_control_amd.controllers import OSC
    c = OSC(ur5.Config(), kp=200)
    for _ in range(100): c.generate(q[0], dq[0], t[0])
    t0 = time.perf_counter()
    for _ in range(2000): c.generate(q[0], dq[0], t[0])
    print(f"  OSC.generate single state through the classes: {(time.perf_counter()-t0)/2000*1e6:.2f} us/call")
    pr = cProfile.Profile(); pr.enable()
    for _ in range(2000): c.generate(q[0], dq[0], t[0])
    pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(12)
PY
export PYTHONPATH="$GRAFT_REPO_ROOT"
for m in 0 1048576 16777216; do echo "ABRK_ZEROCOPY_MAX=$m"; ABRK_ZEROCOPY_MAX=$m python /tmp/staged.py; done
python /tmp/staged.py prof 2>&1 | tail -40
