"""One process, one control loop, every GPU of the node: shards that LIVE on the devices, and several independent
loops merged into one launch on one GPU.

    python examples/one_process_every_gpu.py        (needs an MI355X; with one GPU every shard maps to device 0)

The reference evaluates one state per Python call (examples/PyGame/force_osc_xy.py:57-78: feedback -> ctrlr.generate ->
send_forces); here the states of a whole fleet are scattered over the devices once, every tick only enqueues work on each
device, and the host reads torques back when it wants them.
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # run from a checkout

import abr_control_amd as abrk
from abr_control_amd import _abi, engine
from abr_control_amd.arms import ur5
from abr_control_amd.controllers import OSC, Damping
from abr_control_amd.sharding import MultiDevice

rc = ur5.Config()
mk = lambda: OSC(rc, kp=200, ki=0.1, null_controllers=[Damping(rc, kv=10)])
rng = np.random.RandomState(1)
B = 1 << 18
Q, DQ, T = rng.uniform(0, 2 * np.pi, (B, 6)), rng.uniform(0, 5, (B, 6)), rng.uniform(-1, 1, (B, 6))

n_dev = abrk.device_count()
md = MultiDevice(list(range(n_dev)) if n_dev > 1 else [0] * 8)     # one GPU: eight shards on it (the same code path)
qs, dqs, ts = md.scatter(Q), md.scatter(DQ), md.scatter(T)           # scattered ONCE: the shards stay on their devices
ctrlr = mk()
us = md.generate(ctrlr, qs, dqs, ts)                                 # only enqueues; -> ShardedArray
md.sync()
assert np.array_equal(us.numpy(), mk().generate(Q, DQ, T))           # bit-equal to the unsharded call
print(f"{B} states as {len(md.devices)} resident shards on {n_dev} device(s): one call, results gathered on demand")

# the per-tick loop: one recorded plan per shard, K ticks on every device from ONE call
ctrlr = mk()
plan = md.record_generate(ctrlr, qs, dqs, ts)
plan.launch_graph(50)
plan.sync()
t0 = time.perf_counter()
plan.launch_graph(50)
plan.sync()
dt = (time.perf_counter() - t0) / 50
print(f"  {dt * 1e6:.1f} us per tick of all shards = {B / dt / 1e9:.2f} G control steps/s; integral state stays sharded:",
      type(ctrlr.integrated_error).__name__)
Q2 = rng.uniform(0, 2 * np.pi, (B, 6))
qs.copy_from_numpy(Q2)                                               # new joint angles into the same buffers
plan.launch()
plan.sync()
plan.close()

# several INDEPENDENT loops with the same controller on one GPU: streams of their own overlap two at a time
# (profiles/round6/concurrent_streams.md) - merged into one launch they fill the chip
n_loops, rows = 16, 4096
loops = engine.MergedLoops(rc.arm_id, 6, _abi.make_osc_params(6, kp=200), [rows] * n_loops)
for i in range(n_loops):                                             # every loop feeds its own rows (DeviceArray views)
    lo = i * rows
    v = loops.loop(i)
    v.q.copy_from_numpy(Q[lo:lo + rows]), v.dq.copy_from_numpy(DQ[lo:lo + rows]), v.target.copy_from_numpy(T[lo:lo + rows])
loops.launch_graph(100)
loops.stream.sync()
t0 = time.perf_counter()
loops.launch_graph(100)
loops.stream.sync()
dt = (time.perf_counter() - t0) / 100
u3 = loops.loop(3).u.numpy(loops.stream)
assert np.array_equal(u3, OSC(rc, kp=200).generate(Q[3 * rows:4 * rows], DQ[3 * rows:4 * rows], T[3 * rows:4 * rows]))
print(f"{n_loops} loops x {rows} rows merged: {dt * 1e6:.2f} us per tick = {n_loops * rows / dt / 1e9:.1f} G control steps/s")
loops.close()
