"""The N > 1 path on CPU: two processes over gloo, each evaluating its contiguous shard
(bench.py's scheme: independent shards, barrier + MAX-over-ranks timing, no data-path
collective).  The per-shard compute is the hostsim row program (no GPU here)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from abr_control_amd.sharding import shard_range, shard_rows
from tests.conftest import REPO


def test_shard_ranges_cover_exactly():
    for B in (0, 1, 7, 64, 4096, (1 << 20) + 3):
        for world in (1, 2, 3, 8):
            edges = [shard_range(B, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == B
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in edges]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)
    a = np.arange(10).reshape(5, 2)
    assert np.array_equal(shard_rows([a, None], 1, 2)[0], a[3:])


WORKER = r'''
import os, sys, time
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from abr_control_amd import _abi
from abr_control_amd.sharding import dist_env, shard_rows
from tests import hostsim
rank, local_rank, world = dist_env()
dist.init_process_group("gloo", rank=rank, world_size=world)
rng = np.random.RandomState(1)
B = 1001
q, dq, t = rng.uniform(0, 6.28, (B, 6)), rng.uniform(0, 5, (B, 6)), rng.uniform(-1, 1, (B, 6))
p = _abi.make_osc_params(6, kp=200)
qs, dqs, ts = shard_rows([q, dq, t], rank, world)
dist.barrier(); t0 = time.perf_counter()
u = hostsim.osc_generate("ur5", p, qs, dqs, ts)
dist.barrier(); wall = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
dist.all_reduce(wall, op=dist.ReduceOp.MAX)
# verification only (NOT part of the data path): gather shards on rank 0 and compare with the full batch
parts = [None] * world
dist.all_gather_object(parts, u)
if rank == 0:
    full = hostsim.osc_generate("ur5", p, q, dq, t)
    assert np.array_equal(np.concatenate(parts), full)
    print("OK", float(wall[0]) > 0, sum(len(x) for x in parts))
dist.barrier(); dist.destroy_process_group()
'''


def test_two_rank_gloo_shards_reassemble(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    w = tmp_path / "worker.py"
    w.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(w), REPO], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "OK True 1001" in outs[0]


GROUP_WORKER = r'''
import sys, time
sys.path.insert(0, sys.argv[1])
from abr_control_amd.sharding import HostGroup, dist_env
rank, local_rank, world = dist_env()
if rank == 1:
    time.sleep(0.3)  # rank 0 is listening long before: connect retries are for the opposite order, rank 2 below
g = HostGroup(rank, world, timeout=30)
g.barrier()
t0 = time.perf_counter()
if rank == 0:
    time.sleep(0.2)
g.barrier()  # nobody leaves before the slowest rank arrived
waited = time.perf_counter() - t0
assert waited >= 0.19, waited
assert g.max(float(rank) + 0.5) == world - 0.5
got = g.exchange({"rank": rank, "u": [rank] * 3})
assert [x["rank"] for x in got] == list(range(world)) and got[-1]["u"] == [world - 1] * 3
g.close()
print("OK", rank)
'''


def test_three_rank_host_group_barrier_max_gather():
    """bench.py's rank coordination (abr_control_amd.sharding.HostGroup): barrier, max-over-ranks, gather - three
    processes over a local socket, no communication library"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in (2, 1, 0):  # rank 0 last: the others retry until it listens
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="3", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", GROUP_WORKER, REPO], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
        if rank == 2:
            import time

            time.sleep(0.2)
    outs = [p.communicate(timeout=120) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
        assert o.startswith("OK")
