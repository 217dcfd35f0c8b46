#!/usr/bin/env python3
"""Why does the fused "u + Tx, J, M, g" leg (oscF, 840 B per row, HBM-bound) sit on one of two levels per process (0.69 /
0.79 of HBM peak, rounds 3-6)?  Hypothesis: where the eight arrays start relative to each other in the HBM channel
interleave.  This probe carves all of them out of ONE allocation at controlled byte offsets (`pad` between consecutive
arrays) and times the launch for each padding, several times over, in one process."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import abr_control_amd as a  # noqa: E402
from abr_control_amd import _abi, engine  # noqa: E402
from abr_control_amd._lib import DeviceArray, check, lib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4 << 20
n = 6
arm_id = check(lib().abrk_arm_builtin(b"ur5"))
p = _abi.make_osc_params(6, kp=200)
st = a.Stream(0)
shapes = [("q", (B, n)), ("dq", (B, n)), ("t", (B, 6)), ("u", (B, n)), ("Tx", (B, 3)), ("J", (B, 6, n)), ("M", (B, n, n)), ("g", (B, n))]
total = sum(int(np.prod(s)) * 8 for _, s in shapes)
rng = np.random.RandomState(1)
host = {"q": rng.uniform(0, 2 * np.pi, (B, n)), "dq": rng.uniform(0, 5, (B, n)), "t": rng.uniform(-1, 1, (B, 6))}


def view(base, off, shape):
    v = DeviceArray.__new__(DeviceArray)
    v.shape, v.dtype, v.device = tuple(shape), np.dtype(np.float64), 0
    v.nbytes = int(np.prod(shape)) * 8
    v.ptr = base.ptr + off
    v._base = base
    return v


for rep in range(2):
    for pad in (0, 256, 512, 1024, 2048, 4096, 8192, 16384, 65536, 1 << 20, (1 << 20) + 4096, 3 << 19):
        big = DeviceArray((total + len(shapes) * (pad + (2 << 20)),), np.uint8, 0)
        off, arrs = 0, {}
        for k, (name, shp) in enumerate(shapes):
            arrs[name] = view(big, off, shp)
            off += (arrs[name].nbytes + 255) // 256 * 256 + pad
        for name in ("q", "dq", "t"):
            arrs[name].copy_from_numpy(host[name])
        out = {w: arrs[w] for w in ("Tx", "J", "M", "g")}
        go = lambda: engine.osc_generate(arm_id, n, p, arrs["q"], arrs["dq"], arrs["t"], u=arrs["u"], device=0, stream=st,
                                         want=("Tx", "J", "M", "g"), out=out)
        for _ in range(5):
            go()
        st.sync()
        e0, e1 = a.Event(0), a.Event(0)
        e0.record(st)
        for _ in range(40):
            go()
        e1.record(st)
        st.sync()
        us = e1.elapsed_ms_since(e0) / 40 * 1e3
        print(f"rep {rep} pad {pad:8d}: base mod 2MiB {big.ptr % (2 << 20):8d}  {us:8.1f} us  frac {B * 840 / (us * 1e-6) / 8e12:.3f}", flush=True)
        del arrs, out, big
