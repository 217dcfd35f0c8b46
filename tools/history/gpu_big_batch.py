"""Maximum-size check on the GPU box: 2^26 rows (9 GiB of inputs) through the three-row, Coriolis and six-row (deferred
second pass, 4 M-row worklist) OSC kernels in one call each; the rows are a 1 M-row block tiled 64 times, so every block of
the result must equal the 1 M-row call bit for bit.  python tools/gpu_big_batch.py"""
import numpy as np, time, sys
sys.path.insert(0, '.')
import abr_control_amd as a
from abr_control_amd import _abi, engine
from abr_control_amd._lib import check, lib
arm = check(lib().abrk_arm_builtin(b"ur5"))
B = 1 << 26
rng = np.random.RandomState(5)
blk = 1 << 20
qb, dqb, tb = rng.uniform(0, 2*np.pi, (blk, 6)), rng.uniform(0, 5, (blk, 6)), rng.uniform(-1, 1, (blk, 6))
q, dq, t = (np.tile(x, (B // blk, 1)) for x in (qb, dqb, tb))
print("host arrays", q.nbytes * 3 / 2**30, "GiB", flush=True)
s = a.Stream(0)
qd, dd, td = (a.DeviceArray.from_numpy(x) for x in (q, dq, t))
u = a.DeviceArray((B, 6))
for name, P in (("xyz", _abi.make_osc_params(6, kp=200)), ("xyz+C", _abi.make_osc_params(6, kp=200, use_C=True)),
                ("six rows", _abi.make_osc_params(6, kp=100, ko=80, ctrlr_dof=[1]*6))):
    ref = engine.osc_generate(arm, 6, P, qb, dqb, tb)
    engine.osc_generate(arm, 6, P, qd, dd, td, u=u, stream=s); s.sync()
    t0 = time.perf_counter()
    engine.osc_generate(arm, 6, P, qd, dd, td, u=u, stream=s); s.sync()
    dt = time.perf_counter() - t0
    un = u.numpy()
    ok = all(np.array_equal(un[k*blk:(k+1)*blk], ref) for k in (0, 1, 31, 62, 63))
    print(name, "B =", B, "%.1f ms" % (dt*1e3), "%.2f G rows/s" % (B/dt/1e9), "blocks equal to the 1M-row result:", ok, flush=True)
