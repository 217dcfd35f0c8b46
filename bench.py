#!/usr/bin/env python3
"""Benchmark of the hot path: batched OSC.generate on MI355X.

  python bench.py --gpus N --steps K --warmup W

With N > 1 and no WORLD_SIZE in the environment the command LAUNCHES ITSELF: it starts N copies of itself, one process
per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT exported, HIP_VISIBLE_DEVICES untouched), relays rank
0's ONE JSON line and fails if any rank fails; fewer than N devices -> one JSON line {"error": ..., "devices_seen": k},
rc 1.  Under an external one-process-per-GPU launcher that exports the same variables the command is a rank.

One "step" = one pass of the fused OSC kernel over one batch of synthetic joint states that
are already resident in HBM.  Default workload = BASELINE.json configs[1] (UR5 6-DOF OSC,
batch 4096 per GPU, fp64, xyz control + gravity).  With N > 1 every rank evaluates its own
shard (weak scaling, no collective on the data path); rank 0 prints ONE JSON line.

Besides the contract fields the line carries
  roofline        - dominant kernel on an HBM-sized batch (>> 256 MiB Infinity Cache),
                    HIP-event timed on the launch stream: algorithmic bytes / launch time vs 8 TB/s,
                    plus the FP64-VALU view (this kernel is above the FP64 ridge).  `frac` is the SUSTAINED figure:
                    the kernel launched back to back for --sustain-seconds (default 2 s), mean of the last second
                    (the chip is power-limited on these kernels: the first launches after idle run at boost clocks,
                    the limiter overshoots, then the rate settles); the short run is kept beside it as `short_run`;
  roofline_config - the same accounting for the config-sized (cache-resident, launch-bound) batch;
  cpu_baseline    - the CPU oracle (plain C port of the reference path) on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
FP64_VALU_PEAK_TF = 78.6   # = 1/2 of the 157.3 TF fp32 vector peak
# VALU issue peaks at the nominal 2.4 GHz, in wavefront-lane instructions per second: 1024 SIMDs x 16 lanes/clk for
# fp64 (a wave64 v_fma_f64 issues in 4 cycles), x 32 lanes/clk for fp32 (2 cycles; v_pk_* take 4 - no packed gain).
# Measured by tools/microbench/valu_rates.hip on a short boost-clock run: 36.7 T (fp64), 65.1 T (fp32).
ISSUE_PEAK_NOMINAL = {"f64": 1024 * 16 * 2.4e9, "f32": 1024 * 32 * 2.4e9}
# committed rocprofv3 evidence.  (Earlier rounds are not consulted: kernel names changed - a template parameter was added -
# and round 2's traffic.json keyed the grid-stride kernels by grid threads instead of rows.)
PROFILE_DIRS = ("round6", "round5", "round4", "round3")
# the reference's own Cython path timed on a GPU box's host (cpu_baseline fallback where no staged reference travels)
BASELINE_DIRS = ("round6", "round5", "round4", "round3", "round2")

WORKLOADS = {
    # name: (arm, batch per GPU, dtype, kind, params kwargs, algorithmic flops per eval)
    "cfg1": ("twojoint", 1, "f64", "osc", dict(kp=10, kv=3, ctrlr_dof=[1, 1, 0, 0, 0, 0]), 150),
    "cfg2": ("ur5", 4096, "f64", "osc", dict(kp=200), 3500),
    "cfg3": ("jaco2", 16384, "f64", "osc_damp", dict(kp=200), 4500),
    "cfg4": ("ur5", (1 << 20) // 8, "f64", "osc", dict(kp=200, use_g=True, use_C=True), 10000),
    "cfg5": ("threejoint", 65536, "f32", "sliding", dict(), 1200),
    # config 2's law in single precision (every kernel exists in both arithmetic types; tolerance 1e-4 as for config 5)
    "cfg2_f32": ("ur5", 4096, "f32", "osc", dict(kp=200), 3500),
    # not a BASELINE config: the batched robot_config surface (Tx, J, M, g in one launch) - the
    # HBM-bound "full outputs" mode of SURVEY.md 8d
    "dynF": ("ur5", 4096, "f64", "dyn", dict(want=("Tx", "J", "M", "g")), 2500),
    # the two velocity-dependent robot_config functions, the reference's most expensive ones (SURVEY 8a rows a6, a7:
    # C 6338 + dJ 820 operations after CSE, 4.9 us per call): Christoffel matrix C [n,n] and dJ [6,n] from (q, dq)
    "dynC": ("ur5", 4096, "f64", "dyn", dict(want=("C", "dJ")), 7200),
    # SURVEY 8d "Mode F": the control signal AND the robot_config outputs it consumed (Tx, J, M, g of the EE) from ONE
    # launch of the fused kernel (abrk_osc_generate_full_batch) - 192 + 648 = 840 B per row, HBM-bound
    "oscF": ("ur5", 4096, "f64", "osc_full", dict(kp=200), 3500),
    # ... plus the Christoffel matrix C(q, dq) [n,n] (base_config.py:320-336; SURVEY 8d: +288 B per row = 1128 B)
    "oscFC": ("ur5", 4096, "f64", "osc_full", dict(kp=200, want=("Tx", "J", "M", "g", "C")), 10000),
    # SURVEY 8f-1 (first "next" row): the examples' closed loop on the device - OSC.generate followed by the
    # two-link plant step (arms/twojoint/arm_sim.py:101-137), `rollout_steps` control steps per launch
    "rollout": ("twojoint", 4096, "f64", "rollout", dict(kp=20, use_C=True, ctrlr_dof=[1, 1, 0, 0, 0, 0]), 600),
    # InverseKinematics.generate_path, 200 iterations per path inside one launch (method 3)
    "ik": ("ur5", 4096, "f64", "ik", dict(method=3, n_timesteps=200), 2500),
    # position + orientation control (all six task rows, orientation algorithm 0): the masked six-row kernel
    "osc6": ("ur5", 4096, "f64", "osc", dict(kp=200, ko=150, kv=25, ctrlr_dof=[1] * 6), 6000),
    # the reference benchmark's Jaco2 setting (examples/timing_plots.py:37: ctrlr_dof = [True] * 5 + [False]): five of the
    # six task rows on the general (non-orthogonal) chain - the masked six-row kernels of a one-wave-per-SIMD arm
    "osc5_j2": ("jaco2", 4096, "f64", "osc", dict(kp=200, ctrlr_dof=[1] * 5 + [0]), 7000),
    # Sliding on the six-joint Jaco2 (general affine chain, full Christoffel matrix + dJ): the heaviest kernel of the set
    "sliding_j2": ("jaco2", 65536, "f64", "sliding", dict(), 12000),
    # the remaining secondary controllers as their own kernels (u [B,n] each)
    # the Joint controller (joint.py:104-131, gravity-compensated PD in joint space), its own kernel
    "joint": ("ur5", 4096, "f64", "joint", dict(kp=50, kv=7), 2700),
    "limits": ("ur5", 4096, "f64", "limits", dict(), 60),
    "floating": ("ur5", 4096, "f64", "floating", dict(dynamic=True, task_space=True), 3300),
    "obstacles": ("ur5", 4096, "f64", "obstacles", dict(threshold=0.3, gain=30, obstacles=[
        [0.3, 0.2, 0.4, 0.1], [-0.2, 0.4, 0.3, 0.05], [0.1, -0.3, 0.6, 0.15]]), 6000),
}
ROLLOUT_STEPS = 1000
ROOFLINE_WARMUP = 12


DYN_OUT = lambda n: {"Tx": 3, "J": 6 * n, "M": n * n, "g": n, "C": n * n, "dJ": 6 * n}  # values per row


def algorithmic_bytes(n, esz, kind, want=None):
    """SURVEY.md 8d.  Mode U (controllers): read q, dq [n] + target, write u [n].
    Full-output dynamics: read q [n] (+ dq [n] for C, dJ), write the requested outputs (Tx[3], J[6,n], M[n,n], g[n], ...)."""
    if kind == "dyn":
        want = want or ("Tx", "J", "M", "g")
        return esz * n * (2 if ("C" in want or "dJ" in want) else 1) + esz * sum(DYN_OUT(n)[w] for w in want)
    if kind == "osc_full":  # Mode U (q, dq, target in; u out) + the requested robot_config outputs
        want = want or ("Tx", "J", "M", "g")
        return esz * (2 * n + 6) + esz * n + esz * sum(DYN_OUT(n)[w] for w in want)
    if kind == "ik":  # per launch and row: q, target in; position + velocity paths out
        return esz * (n + 6) + esz * 2 * 200 * n
    if kind == "rollout":  # per launch and row: q, dq in/out + target; amortised over ROLLOUT_STEPS
        return esz * (4 * n + 6)
    if kind in ("limits", "obstacles"):  # q in, u out
        return esz * 2 * n
    if kind == "floating":  # q, dq in, u out
        return esz * 3 * n
    if kind == "joint":  # q, dq, target [n] in, u out
        return esz * 4 * n
    nt = 3 if kind == "sliding" else 6
    return esz * (2 * n + nt) + esz * n


def make_inputs(seed, B, n, nt, dt):
    """the reference benchmark's input distribution (examples/timing_plots.py:18-20)"""
    rng = np.random.RandomState(seed)
    q = rng.uniform(0, 2 * np.pi, (B, n)).astype(dt)
    dq = rng.uniform(0, 5, (B, n)).astype(dt)
    t = rng.uniform(-1, 1, (B, nt)).astype(dt)
    return q, dq, t


class Runner:
    """device-resident inputs + one launch per step()"""

    def __init__(self, workload, B, device, stream, global_rows=None):
        """global_rows = (lo, G): this runner holds rows [lo, lo + B) of ONE global batch of G seeded rows (the strong-
        scaling leg: every rank draws the same G rows and keeps its contiguous shard)"""
        import abr_control_amd as a
        from abr_control_amd import _abi, engine
        from abr_control_amd._lib import check, lib

        arm, _, dts, kind, kw, self.flops = WORKLOADS[workload]
        self.a, self.engine, self.kind, self.arm = a, engine, kind, arm
        self.dt = np.float64 if dts == "f64" else np.float32
        tab = _abi.load_table(arm)
        self.n = tab["n_joints"]
        self.nt = 3 if kind == "sliding" else 6
        self.arm_id = check(lib().abrk_arm_builtin(arm.encode()))
        self.B, self.device, self.stream = B, device, stream
        if global_rows is None:
            q, dq, t = make_inputs(1, B, self.n, self.nt, self.dt)
        else:
            lo, G = global_rows
            q, dq, t = (np.ascontiguousarray(x[lo:lo + B]) for x in make_inputs(1, G, self.n, self.nt, self.dt))
        self.host = (q, dq, t)
        self.q = a.DeviceArray.from_numpy(q, device)
        self.dq = a.DeviceArray.from_numpy(dq, device)
        self.t = a.DeviceArray.from_numpy(t, device)
        self.u = a.DeviceArray((B, self.n), self.dt, device)
        if kind == "ik":
            self.params = _abi.make_ik_params(**kw)
            T = kw["n_timesteps"]
            self.ik_out = (a.DeviceArray((B, T, self.n), self.dt, device), a.DeviceArray((B, T, self.n), self.dt, device))
        elif kind == "rollout":
            rc_L = np.array([[0, 0, 0], [0, 0, 0], [1.0, 0, 0], [1.0, 0, 0], [0.6, 0, 0], [0.6, 0, 0]])
            M = [np.diag(tab["mdiag"][l]) for l in range(3)]
            self.plant = _abi.make_twolink_plant(rc_L, M, 0.001)
            nulls = [_abi.make_damping(10), _abi.make_resting([np.pi / 4, np.pi], kp=50, kv=np.sqrt(50))]
            self.params = _abi.make_osc_params(2, null_controllers=nulls, **kw)
            q0 = (np.array([np.pi / 4, np.pi / 4]) + np.random.RandomState(1).uniform(-0.6, 0.6, (B, 2))).astype(self.dt)
            t6 = np.zeros((B, 6), self.dt)
            t6[:, :2] = np.random.RandomState(2).uniform(-1.5, 1.5, (B, 2))
            self.q0 = q0
            self.q = a.DeviceArray.from_numpy(q0, device)
            self.dq = a.DeviceArray.from_numpy(np.zeros((B, 2), self.dt), device)
            self.t = a.DeviceArray.from_numpy(t6, device)
        elif kind in ("dyn", "osc_full"):
            n = self.n
            shapes = {"Tx": (3,), "J": (6, n), "M": (n, n), "g": (n,), "C": (n, n), "dJ": (6, n)}
            self.want = kw.get("want", ("Tx", "J", "M", "g"))
            self.dyn_out = {w: a.DeviceArray((B,) + shapes[w], self.dt, device) for w in self.want}
            if kind == "osc_full":
                self.params = _abi.make_osc_params(self.n, **{k: v for k, v in kw.items() if k != "want"})
        elif kind == "sliding":
            self.params = _abi.make_sliding_params(self.n)
        elif kind == "limits":
            self.params = _abi.make_limits_params(
                self.n, [np.pi / 5, 1.0, 5.5, None, 0.4, 2.0], [np.pi / 2, 4.0, 0.8, None, None, 2.5],
                [100.0, 7.5, 3.0, 1.0, 4.0, 2.0], [False, False, True, False, False, False],
                [False, True, False, False, False, True])
        elif kind == "floating":
            self.params = kw
        elif kind == "joint":
            self.params = _abi.make_joint(**kw)
            self.tj = a.DeviceArray.from_numpy(np.random.RandomState(3).uniform(0, 2 * np.pi, (B, self.n)).astype(self.dt), device)
        elif kind == "obstacles":
            self.params = _abi.make_obstacles_params(**kw)
        else:
            nulls = [_abi.make_damping(10)] if kind == "osc_damp" else []
            self.params = _abi.make_osc_params(self.n, null_controllers=nulls, **kw)
        self.plan = None
        # launch-bound at the config batch: the K steps are replayed as hipGraph launches of 100 kernel nodes each
        # (abrk_plan_launch_graph; same kernel, same buffers, one node per step).  ABRK_BENCH_GRAPH=0: one
        # hipLaunchKernel per step (4.55 instead of 4.06 us per step at B = 4096)
        self.graph_steps = int(os.environ.get("ABRK_BENCH_GRAPH", "100"))
        if kind not in ("rollout", "ik"):
            # the per-tick launch of a control loop on fixed device buffers: the call is recorded once (arguments
            # validated and converted, abrk_plan_begin/end); a step then only enqueues the kernel.  (The rollout and
            # the IK paths already run hundreds of iterations per launch.)
            with engine.Plan(device, stream) as self.plan:
                self._enqueue()
        if kind != "rollout":
            # one untimed launch at construction: the first launch of a kernel loads its code object (milliseconds) -
            # initialisation, not a step, whatever --warmup says (the rollout advances its state in place: left alone)
            self.step()
            stream.sync()
        self.bytes_per_eval = algorithmic_bytes(self.n, np.dtype(self.dt).itemsize, kind,
                                                getattr(self, "want", None) if kind in ("dyn", "osc_full") else None)
        self.evals_per_launch = B * (ROLLOUT_STEPS if kind == "rollout" else kw["n_timesteps"] if kind == "ik" else 1)

    def kernel_name(self):
        """the instantiation this workload launches, spelled as rocprofv3 prints it (key of profiles/*/traffic.json)"""
        t = "double" if self.dt == np.float64 else "float"
        arm = f"abrk::StaticArm<abrk::Tab_{self.arm}>"
        k = self.kind
        if k in ("osc", "osc_damp"):
            p = self.params
            dof = list(p.ctrlr_dof)
            fast = dof == [1, 1, 1, 0, 0, 0] and p.ref_frame == 2 * self.n + 1
            b = lambda v: "true" if v else "false"
            # six task rows: first pass (PASS = 1, the dominant kernel) + a second pass for the rows whose pseudo-inverse
            # truncates - 64 ... 65536 rows the finish kernel on hand-over records (osc6_finish_kernel), beyond that the
            # complete row program once more (PASS = 0); ABRK_MEASUREMENT=1 ABRK_NO_HANDOVER=1: the round-3 scheme (inline
            # below 16 k rows)
            no_handover = os.environ.get("ABRK_MEASUREMENT") == "1" and os.environ.get("ABRK_NO_HANDOVER")
            six_two_pass = (not fast) and (self.B >= 16384 or not no_handover)
            km = 3 if fast else (2 if dof == [1, 1, 0, 0, 0, 0] and self.n <= 3 and p.ref_frame == 2 * self.n + 1 else 6)
            # ... and, the bench never asking for the training signal, the plain law's first pass is the NOTS variant
            nots = six_two_pass and not p.n_null and not os.environ.get("ABRK_BENCH_TS")
            # ... and, the reference frame being the end effector, the first pass of the plain law is the EEF instantiation
            # (round 6: no frame capture in the forward kinematics)
            eef = (six_two_pass and not p.n_null and p.ref_frame == 2 * self.n + 1
                   and self.arm in ("ur5", "threejoint", "twojoint", "onejoint"))  # orthogonal chains: abrk_kernels.h kEefBuilt
            return (f"osc_kernel<{arm}, {t}, {km}, {b(p.use_C)}, {1 if p.n_null else 0}, {1 if six_two_pass else 0}, "
                    f"{b(nots)}, {b(eef)}>")
        if k == "dyn":
            return f"dyn_kernel<{arm}, {t}, {'true' if ('C' in self.want or 'dJ' in self.want) else 'false'}>"
        if k == "osc_full":
            vel = "C" in self.want or "dJ" in self.want
            return (f"osc_full_kernel<{arm}, {t}, 3, {'true' if self.params.use_C else 'false'}, {2 if vel else 0}, "
                    f"{'true' if vel else 'false'}>")
        if k == "limits":
            return f"limits_kernel<{self.n}, {t}>"
        if k == "rollout":
            return f"rollout_kernel<{arm}, {t}, {'true' if self.params.use_C else 'false'}>"
        obs_plain = os.environ.get("ABRK_MEASUREMENT") == "1" and os.environ.get("ABRK_OBS_PLAIN")
        if k == "obstacles" and self.arm in ("ur5", "threejoint") and not obs_plain:
            return f"obstacles_lds_kernel<{arm}, {t}>"  # orthogonal chains: heavy pairs redistributed through LDS
        return f"{k}_kernel<{arm}, {t}>"

    def grid_threads(self):
        """threads of the dominant kernel's launch (grid x 64), as rocprofv3's Grid_Size prints it: the grid-stride
        kernels cap their grid, so rows != threads for them (abrk_kernels.h: kSlidingMaxBlocks, kObstaclesMaxBlocks,
        kKm6GridCap) - tools/summarize_profiles.py maps (kernel, Grid_Size) back to rows through this"""
        blocks = (self.B + 63) // 64
        if self.kind == "sliding":
            blocks = min(blocks, 256 * 32 * 4)
        elif self.kind == "obstacles":
            blocks = min(blocks, 4096)
        elif (self.kind in ("osc", "osc_damp") and ", 6, " in self.kernel_name() and self.arm == "jaco2"
              and (self.params.use_C or self.params.n_null)):
            # the six-row first pass of a general chain with the Coriolis vector or fused null controllers (one wave per
            # SIMD) is a persistent grid; the plain law's runs one row per lane since round 6 (two waves per SIMD)
            blocks = min(blocks, 4096)
        return blocks * 64

    def step(self):
        if self.plan is not None:
            self.plan.launch()
        else:
            self._enqueue()

    def sustained(self, seconds, est_ms):
        """the kernel launched back to back for >= `seconds`: HIP events every `chunk` launches, no host sync in
        between (the stream never runs dry) -> per-launch ms of every chunk, in time order.  est_ms: the short run's
        per-launch time (sizes the run)."""
        a = self.a
        chunk = max(1, min(64, int(20.0 / max(est_ms, 1e-3))))          # ~20 ms of GPU time per chunk
        n_chunks = max(4, int(np.ceil(seconds * 1e3 / (chunk * est_ms) * 1.15)))  # the limiter slows the run down
        # HBM-sized launches (hundreds of microseconds each) need no graph to keep the stream fed: plain launches from the
        # recorded plan; only launch-bound kernels (< 50 us) are chunked into hipGraph replays
        graph = self.plan is not None and chunk > 1 and est_ms < 0.05
        if graph:
            self.plan.launch_graph(chunk)  # builds the graph (untimed)
            self.stream.sync()
        evs = [a.Event(self.device) for _ in range(n_chunks + 1)]
        evs[0].record(self.stream)
        for i in range(n_chunks):
            if graph:
                self.plan.launch_graph(chunk)
            else:
                for _ in range(chunk):
                    self.step()
            evs[i + 1].record(self.stream)
        self.stream.sync()
        return [evs[i + 1].elapsed_ms_since(evs[i]) / chunk for i in range(n_chunks)], chunk

    def _enqueue(self):
        if self.kind == "ik":
            self.engine.ik_generate_path(self.arm_id, self.n, self.params, self.q, self.t, dtype=self.dt,
                                         device=self.device, stream=self.stream, position_path=self.ik_out[0],
                                         velocity_path=self.ik_out[1])
        elif self.kind == "osc_full":
            self.engine.osc_generate(self.arm_id, self.n, self.params, self.q, self.dq, self.t, u=self.u,
                                     dtype=self.dt, device=self.device, stream=self.stream, want=self.want,
                                     out=self.dyn_out)
        elif self.kind == "rollout":
            self.engine.osc_rollout_twolink(self.arm_id, self.params, self.plant, self.q, self.dq, self.t,
                                            ROLLOUT_STEPS, dtype=self.dt, device=self.device, stream=self.stream)
        elif self.kind == "dyn":
            vel = self.dq if ("C" in self.want or "dJ" in self.want) else None
            self.engine.dynamics(self.arm_id, self.n, self.q, vel, None, None, self.want, self.dt, self.device,
                                 self.stream, out=self.dyn_out)
        elif self.kind == "sliding":
            self.engine.sliding_generate(self.arm_id, self.n, self.params, self.q, self.dq, self.t, u=self.u,
                                         dtype=self.dt, device=self.device, stream=self.stream)
        elif self.kind == "limits":
            self.engine.avoid_joint_limits_generate(self.n, self.params, self.q, u=self.u, dtype=self.dt,
                                                    device=self.device, stream=self.stream)
        elif self.kind == "floating":
            self.engine.floating_generate(self.arm_id, self.n, self.params["dynamic"], self.params["task_space"],
                                          self.q, self.dq, u=self.u, dtype=self.dt, device=self.device,
                                          stream=self.stream)
        elif self.kind == "joint":
            self.engine.joint_generate(self.arm_id, self.n, self.params, True, self.q, self.dq, self.tj, None, u=self.u,
                                       dtype=self.dt, device=self.device, stream=self.stream)
        elif self.kind == "obstacles":
            self.engine.avoid_obstacles_generate(self.arm_id, self.n, self.params, self.q, u=self.u, dtype=self.dt,
                                                 device=self.device, stream=self.stream)
        elif os.environ.get("ABRK_BENCH_TS"):
            # measurement switch: also ask for the training signal (osc.py:297) - what the Python OSC class always does;
            # + 8 n bytes per row of output, and the six-row first pass is then the instantiation that carries it
            if not hasattr(self, "ts"):
                self.ts = self.a.DeviceArray((self.B, self.n), self.dt, self.device)
            self.engine.osc_generate(self.arm_id, self.n, self.params, self.q, self.dq, self.t, u=self.u,
                                     training_signal=self.ts, dtype=self.dt, device=self.device, stream=self.stream)
        else:
            self.engine.osc_generate(self.arm_id, self.n, self.params, self.q, self.dq, self.t, u=self.u,
                                     dtype=self.dt, device=self.device, stream=self.stream)

    def timed(self, steps, warmup, barrier=None):
        """-> (wall seconds of the timed region, mean kernel ms per launch from HIP events)"""
        a = self.a
        if self.plan is not None and 8 <= steps < self.graph_steps:
            self.graph_steps = steps  # a short run is one graph of exactly K nodes
        graph = self.used_graph = self.plan is not None and self.graph_steps > 1 and steps >= self.graph_steps
        # measurement switch (tools/history/gpu_latency_r3.sh): a short run as `head` plain launches followed by ONE graph of the
        # remaining K - head nodes - does the GPU-side lead-in of a graph launch hide behind kernels already running?
        head = int(os.environ.get("ABRK_BENCH_HEAD", "0")) if graph and steps == self.graph_steps else 0
        head = min(head, max(steps - 2, 0))
        repeat_c = self.plan is not None and os.environ.get("ABRK_BENCH_REPEAT_C") == "1"
        for _ in range(warmup):
            self.step()
        if graph:  # builds (captures + instantiates) the graph outside the timed region; untimed extra steps
            self.plan.launch_graph(self.graph_steps - head)
        self.stream.sync()
        ev0, ev1 = a.Event(self.device), a.Event(self.device)

        def k_steps():
            if repeat_c:
                self.plan.launch_repeat(steps)  # K plain launches enqueued by ONE C call (abrk_plan_launch_repeat)
            elif graph and head:
                for _ in range(head):
                    self.step()
                self.plan.launch_graph(steps - head)
            elif graph:
                # K steps as ceil(K/G) hipGraph launches of G kernel nodes each (+ a remainder of plain launches)
                for _ in range(steps // self.graph_steps):
                    self.plan.launch_graph(self.graph_steps)
                for _ in range(steps % self.graph_steps):
                    self.step()
            else:
                for _ in range(steps):
                    self.step()

        if barrier:
            barrier()
        self.stream.sync()
        t0 = time.perf_counter()
        k_steps()
        self.stream.sync()
        # this rank's K steps, device-synchronised on both sides; the closing barrier keeps the ranks together but
        # its own (gloo, TCP) latency is not step time - main() takes the MAX of `wall` over the ranks
        wall = time.perf_counter() - t0
        if barrier:
            barrier()
        # the HIP-event time per launch (the `roofline*` blocks) comes from one more, untimed replay of the same K
        # steps: recording the two events inside the wall-clocked region costs ~5 us of host time, which at K = 20
        # config-sized steps is 5 % of the run
        ev0.record(self.stream)
        k_steps()
        ev1.record(self.stream)
        self.stream.sync()
        return wall, ev1.elapsed_ms_since(ev0) / steps


def concurrent_streams_rate(workload, B, device, n_streams, steps, graph_steps=100):
    """Several INDEPENDENT batches of the config size, each with its own stream, launch plan and buffers (e.g. separate
    fleets with separate control loops): the config-sized step is issue-latency-bound on 64 of the chip's 1024 SIMDs, so
    concurrent loops fill the rest.  Reported beside `value` (one loop), never as `value`."""
    import abr_control_amd as a

    streams = [a.Stream(device) for _ in range(n_streams)]
    runs = [Runner(workload, B, device, st) for st in streams]
    reps = max(steps // graph_steps, 1)
    for r in runs:  # warm-up: builds each plan's graph
        r.plan.launch_graph(graph_steps)
    for st in streams:
        st.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        for r in runs:
            r.plan.launch_graph(graph_steps)
    for st in streams:
        st.sync()
    wall = time.perf_counter() - t0
    total_steps = reps * graph_steps
    return {"streams": n_streams, "batch_per_stream": B, "steps_per_stream": total_steps,
            "evals_per_s": round(n_streams * B * total_steps / wall, 1),
            "us_per_step_per_stream": round(wall / total_steps * 1e6, 3)}


def kernel_sources_hash():
    """md5 over the kernel and host sources (abr_control_amd/csrc/*.{h,hip,cpp}, include/*.h): the profile summaries carry
    the hash of the tree they were taken on (`_sources`), the bench line the hash of the tree it runs on - equal hashes
    mean the profiled kernels ARE this tree's kernels, whatever documents were committed in between"""
    import glob
    import hashlib

    h = hashlib.md5()
    files = sorted(glob.glob(os.path.join(REPO, "abr_control_amd", "csrc", "*.h")) + glob.glob(os.path.join(REPO, "abr_control_amd", "csrc", "*.hip"))
                   + glob.glob(os.path.join(REPO, "abr_control_amd", "csrc", "*.cpp")) + glob.glob(os.path.join(REPO, "include", "*.h")))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def _profiled(fname, kernel, batch):
    """(entry, "profiles/<round>/<fname>", commit the profile was taken at) of the committed rocprofv3 evidence for this
    (kernel, rows per launch) - or (None, None, None)"""
    kn = kernel.replace(' ', '')
    key = f"{kn}:{batch}"
    for rnd in PROFILE_DIRS:
        path = os.path.join(REPO, "profiles", rnd, fname)
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        if key in d:
            return dict(d[key], rows=batch), f"profiles/{rnd}/{fname}", _stamp(d)
        # the same kernel profiled at another HBM-sized batch: per-row figures carry over (the caller rescales)
        same = sorted(((int(k.rsplit(":", 1)[1]), k) for k in d if k.startswith(kn + ":")), reverse=True)
        if same and same[0][0] >= (1 << 20) and batch >= (1 << 20):
            return dict(d[same[0][1]], rows=same[0][0]), f"profiles/{rnd}/{fname} (profiled at {same[0][0]} rows)", _stamp(d)
    return None, None, None


def _stamp(d):
    """commit of a profile summary + whether its kernel sources are this tree's"""
    src = d.get("_sources")
    if not src:
        return d.get("_commit")
    same = src == kernel_sources_hash()
    return f"{d.get('_commit')} (kernel sources {src}: {'identical to' if same else 'DIFFERENT from'} this tree's)"


def profiled_traffic(kernel, batch):
    """HBM bytes per launch from the committed rocprofv3 PMC passes of this same command (tools/gpu_profiles_r6.sh ->
    profiles/<round>/traffic.json): NOT a measurement of the run that prints it - the profile's commit is stamped"""
    t, src, commit = _profiled("traffic.json", kernel, batch)
    if t is None:
        return None, None, None
    return round((t["read_bytes"] + t["write_bytes"]) * batch / t["rows"], 1), src, commit


def sustained_stats(per_launch_ms, chunk):
    """mean per-launch time over the LAST SECOND of a back-to-back run (+ the run's course)"""
    t = np.cumsum(np.asarray(per_launch_ms) * chunk)  # ms at the end of each chunk
    total = float(t[-1])
    tail = [x for x, end in zip(per_launch_ms, t) if end > total - 1000.0]
    first = [x for x, end in zip(per_launch_ms, t) if end <= 1000.0] or per_launch_ms[:1]
    return {"seconds": round(total / 1e3, 3), "launches": int(len(per_launch_ms) * chunk), "launches_per_event": chunk,
            "us_per_launch_last_second": round(float(np.mean(tail)) * 1e3, 3),
            "us_per_launch_first_second": round(float(np.mean(first)) * 1e3, 3),
            "us_per_launch_min_chunk": round(float(np.min(per_launch_ms)) * 1e3, 3),
            "us_per_launch_max_chunk": round(float(np.max(per_launch_ms)) * 1e3, 3)}


def roofline_leg(runner, label, steps, sustain_s, barrier=None):
    """one HBM-sized leg: ROOFLINE_WARMUP launches, `steps` timed launches (HIP events), then - unless sustain_s is
    0 - the back-to-back run whose last second gives `frac`"""
    _, ms_short = runner.timed(steps, ROOFLINE_WARMUP, barrier)
    if sustain_s <= 0:
        return roofline(runner, ms_short, label)
    per, chunk = runner.sustained(sustain_s, ms_short)
    st = sustained_stats(per, chunk)
    out = roofline(runner, st["us_per_launch_last_second"] * 1e-3, label)
    out["protocol"] = (f"back-to-back launches for {st['seconds']} s after {ROOFLINE_WARMUP} warm-up + {steps} short-run "
                       f"launches; achieved / frac / us_per_launch = mean of the last second (HIP events every "
                       f"{chunk} launches on the launch stream)")
    out["sustained"] = st
    short = roofline(runner, ms_short, label)
    out["short_run"] = {"launches": steps, "us_per_launch": short["us_per_launch"], "achieved": short["achieved"],
                        "frac": short["frac"]}
    return out


def roofline(runner, ms_per_launch, label):
    evals_s = runner.evals_per_launch / (ms_per_launch * 1e-3)
    gbs = runner.B / (ms_per_launch * 1e-3) * runner.bytes_per_eval / 1e9
    kname = runner.kernel_name()
    traffic, tsrc, tcommit = profiled_traffic(kname, runner.B)
    dts = "f64" if runner.dt == np.float64 else "f32"
    out = {
        "kernel": kname, "grid_threads": runner.grid_threads(),
        "workload": label, "batch": runner.B, "bound": "hbm", "achieved": round(gbs, 3), "peak": HBM_PEAK_GBS,
        "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 5), "traffic": traffic,
        "traffic_source": None if traffic is None else f"{tsrc} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; "
                                                       f"FETCH_SIZE x2 on gfx950), profiled at commit {tcommit}",
        "algorithmic_bytes_per_launch": runner.B * runner.bytes_per_eval,
        "bytes_per_eval": runner.bytes_per_eval, "us_per_launch": round(ms_per_launch * 1e3, 3),
        "evals_per_s": round(evals_s, 1),
    }
    # the VALU view from EXECUTED instructions (SQ_INSTS_VALU / row of the committed PMC pass), not from a
    # reference-derived flop count: lane-instructions per second against the issue peak at the nominal clock
    # (HBM-sized legs only: the PMC pass of a config-sized batch is one cold dispatch, and its cycle counters divided by
    # a replayed kernel's time are not a clock - round 4's line printed 5.27 "GHz" that way)
    c, csrc, ccommit = _profiled("counters.json", kname, runner.B) if runner.B >= (1 << 20) else (None, None, None)
    if c is not None:
        lane_instr_s = evals_s * c["valu_per_row"]
        peak = ISSUE_PEAK_NOMINAL[dts]
        out["valu"] = {
            "binding": f"{dts}_valu_issue", "valu_instr_per_row": round(c["valu_per_row"], 1),
            "salu_instr_per_row": round(c["salu_per_row"], 1),
            "lane_instr_per_s": round(lane_instr_s, 1), "issue_peak_nominal_2p4GHz": peak,
            "frac_of_nominal_issue_peak": round(lane_instr_s / peak, 4),
            # an fp64 instruction retires at most one fma per lane: executed flops <= 2 x instructions
            "executed_tflops_upper_bound": round(2 * lane_instr_s / 1e12, 2),
            "issue_util_profiled": c.get("issue_util"), "clock_ghz_profiled": c.get("clock_ghz"),
            "source": f"{csrc}, profiled at commit {ccommit}; rate of THIS run x profiled instructions per row",
        }
    out["algorithmic_flops_per_eval_reference_cse"] = runner.flops  # SURVEY 8d (SymPy cse of the reference's expressions)
    return out


def parity_vs_reference(device):
    """max |du| of the GPU result against outputs of the reference itself (tests/golden/ur5.npz, written by
    oracle/gen_golden.py): `cfg2_uS` = the reference's Cython path as shipped (float32-rounding wrappers),
    `cfg2_uD` = the same osc.py formulas fed by its fp64 generated functions."""
    from abr_control_amd.arms import ur5
    from abr_control_amd.controllers import OSC

    g = np.load(os.path.join(REPO, "tests", "golden", "ur5.npz"))
    u = OSC(ur5.Config(device=device), kp=200).generate(g["cfg2_q"], g["cfg2_dq"], g["cfg2_target"])
    rel = lambda a, b: np.max(np.abs(a - b), axis=1) / np.max(np.abs(b), axis=1)
    rS, rD = rel(u, g["cfg2_uS"]), rel(u, g["cfg2_uD"])
    return {"rows": int(len(u)), "max_abs_du_vs_cython_ref": float(np.max(np.abs(u - g["cfg2_uS"]))),
            "median_rel_vs_cython_ref": float(np.median(rS)), "max_rel_vs_cython_ref": float(rS.max()),
            "max_abs_du_vs_fp64_ref": float(np.max(np.abs(u - g["cfg2_uD"]))), "max_rel_vs_fp64_ref": float(rD.max()),
            "note": "the shipped reference rounds J,M,g to float32 (base_config.py:223-285): its own distance from "
                    "the fp64 evaluation of the same formulas is the median/max rel vs_cython_ref seen here"}


def host_staged_rate(device):
    """End-to-end rate when the boundary hands over HOST arrays (NumPy in, NumPy out): the library stages
    q, dq, target over PCIe and copies u back on every call.  Reported beside `value`, never as `value`
    (bounded by 63 GB/s / 192 B = 0.33 G evals/s)."""
    from abr_control_amd import _abi, engine
    from abr_control_amd._lib import check, lib

    arm_id = check(lib().abrk_arm_builtin(b"ur5"))
    p = _abi.make_osc_params(6, kp=200)
    res = {}
    for B, reps in ((1, 200), (4096, 100), (1 << 20, 5)):
        q, dq, t = make_inputs(1, B, 6, 6, np.float64)
        u = np.empty((B, 6))
        engine.osc_generate(arm_id, 6, p, q, dq, t, u=u, device=device)
        t0 = time.perf_counter()
        for _ in range(reps):
            engine.osc_generate(arm_id, 6, p, q, dq, t, u=u, device=device)
        dt = (time.perf_counter() - t0) / reps
        res[f"batch_{B}"] = {"us_per_call": round(dt * 1e6, 2), "evals_per_s": round(B / dt, 1)}
    return res


def usable_cores():
    """host cores this process may actually use: the scheduler affinity, capped by the cgroup CPU quota (the GPU
    boxes show 256 logical CPUs but grant a container 16 of them - more threads than that only thrash)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(workload, budget_s=12.0):
    """the oracle (plain-C port of the reference path) on one host core, bounded sample"""
    from abr_control_amd import _abi
    from oracle.oracle import Oracle

    arm, _, _, kind, kw, _ = WORKLOADS[workload]
    o = Oracle(_abi.load_table(arm))
    nt = 3 if kind == "sliding" else 6
    Bs, scale = 2048, 1
    q, dq, t = make_inputs(1, Bs, o.n, nt, np.float64)
    if kind == "dyn":
        want = kw.get("want", ("Tx", "J", "M", "g"))
        f1 = {"Tx": lambda i: o.Tx("EE", q[i]), "J": lambda i: o.J("EE", q[i]), "M": lambda i: o.M(q[i]),
              "g": lambda i: o.g(q[i]), "C": lambda i: o.C(q[i], dq[i]), "dJ": lambda i: o.dJ("EE", q[i], dq[i])}
        fn = lambda: [[f1[w](i) for w in want] for i in range(Bs)]
    elif kind == "sliding":
        p = _abi.make_sliding_params(o.n)
        fn = lambda: o.sliding_batch(p, q, dq, t)
    elif kind == "ik":
        from oracle.oracle import ik_paths

        Bs = 8
        p = _abi.make_ik_params(**kw)
        tab = _abi.load_table(arm)
        fn = lambda: ik_paths(tab, p, q[:Bs], t[:Bs])
        scale = kw["n_timesteps"]  # evals = iterations
    elif kind == "rollout":
        from oracle.oracle import rollout_twolink

        Bs = 8
        tab = _abi.load_table(arm)
        L = np.array([[0, 0, 0], [0, 0, 0], [1.0, 0, 0], [1.0, 0, 0], [0.6, 0, 0], [0.6, 0, 0]])
        plant = _abi.make_twolink_plant(L, [np.diag(tab["mdiag"][l]) for l in range(3)], 0.001)
        nulls = [_abi.make_damping(10), _abi.make_resting([np.pi / 4, np.pi], kp=50, kv=np.sqrt(50))]
        p = _abi.make_osc_params(2, null_controllers=nulls, **kw)
        fn = lambda: rollout_twolink(tab, p, plant, q[:Bs], np.zeros((Bs, 2)), t[:Bs], ROLLOUT_STEPS, ROLLOUT_STEPS)
        scale = ROLLOUT_STEPS  # evals = control steps
    elif kind == "limits":
        from oracle.oracle import avoid_joint_limits_batch

        p = _abi.make_limits_params(o.n, [np.pi / 5, 1.0, 5.5, None, 0.4, 2.0], [np.pi / 2, 4.0, 0.8, None, None, 2.5],
                                    [100.0, 7.5, 3.0, 1.0, 4.0, 2.0], [False, False, True, False, False, False],
                                    [False, True, False, False, False, True])
        fn = lambda: avoid_joint_limits_batch(o.n, p, q)
    elif kind == "floating":
        fn = lambda: o.floating_batch(kw["dynamic"], kw["task_space"], q, dq)
    elif kind == "joint":
        tj = np.random.RandomState(3).uniform(0, 2 * np.pi, q.shape)
        fn = lambda: o.joint_batch(_abi.make_joint(**kw), True, q, dq, tj)
    elif kind == "obstacles":
        p = _abi.make_obstacles_params(**kw)
        fn = lambda: o.avoid_obstacles_batch(p, q)
    elif kind == "osc_full":
        p = _abi.make_osc_params(o.n, **{k: v for k, v in kw.items() if k != "want"})
        withC = "C" in kw.get("want", ())
        fn = lambda: (o.osc_batch(p, q, dq, t), [(o.Tx("EE", q[i]), o.J("EE", q[i]), o.M(q[i]), o.g(q[i])) +
                                                 ((o.C(q[i], dq[i]),) if withC else ()) for i in range(Bs)])
    else:
        nulls = [_abi.make_damping(10)] if kind == "osc_damp" else []
        p = _abi.make_osc_params(o.n, null_controllers=nulls, **kw)
        fn = lambda: o.osc_batch(p, q, dq, t)
    fn()
    reps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s / 2:
        fn()
        reps += 1
    dt1 = time.perf_counter() - t0
    one = reps * Bs * scale / dt1
    # all host cores: the C oracle is called through ctypes (GIL released), one thread per core, each on
    # its own copy of the sample - the reference itself has no multi-core path
    from concurrent.futures import ThreadPoolExecutor

    cores = usable_cores()
    stop = time.perf_counter() + budget_s / 2

    def worker(_):
        n = 0
        while time.perf_counter() < stop:
            fn()
            n += 1
        return n

    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        total = sum(ex.map(worker, range(cores)))
    dtc = time.perf_counter() - t0
    # THE REFERENCE'S OWN Cython path on this box's host cores (north_star: "timed on the host cores of the same box, core
    # count stated"): the reference travels as oracle/_ref/abr_control_ref.tar.gz (oracle/stage_reference.py, packed in
    # the build container; /root/reference does not exist here).  ~4 s on one core + ~4 s on all granted cores.  Without
    # the archive (or sympy / Cython): the figures measured in the build container, labelled as such.
    ref = None
    if workload in ("cfg1", "cfg2", "cfg3", "cfg4", "cfg5"):
        try:
            from oracle import time_reference

            rj = time_reference.measure_staged([workload], cores, budget=4.0)
        except Exception as e:  # noqa: BLE001 - a reported baseline must not take the bench line down
            rj = None
            print(f"cpu_baseline: the staged reference did not run here: {e}", file=sys.stderr)
        if rj is None:
            for rnd in BASELINE_DIRS:
                try:
                    rj = json.load(open(os.path.join(REPO, "profiles", rnd, "reference_cython_baseline.json")))
                    rj["measured_on"] += " [committed figures: no staged reference on this machine]"
                    break
                except (OSError, ValueError, KeyError):
                    rj = None
        if rj and workload in rj["workloads"]:
            ref = dict(rj["workloads"][workload], cores=rj["cores"], measured_on=rj["measured_on"], what=rj["what"],
                       script=rj["script"])
    port = {"value": round(total * Bs * scale / dtc, 1), "unit": "evals/s", "cores": cores, "kind": "port",
            "value_1core": round(one, 1),
            "sample": f"oracle/abrk_oracle.c (plain-C port of the reference path) on seeded rows of the same "
                      f"workload: {reps} x {Bs} rows{f' x {scale} steps' if scale > 1 else ''} on 1 thread in {dt1:.1f} s; "
                      f"{total} x {Bs} rows on {cores} threads in {dtc:.1f} s"}
    if ref is not None and "committed figures" not in ref["measured_on"]:
        # the reference itself ran here: it is the baseline; the compiled port is quoted beside it
        return {"value": ref["evals_per_s_allcores"], "unit": "evals/s", "cores": ref["cores"], "kind": "reference",
                "value_1core": ref["evals_per_s_1core"], "us_per_eval_1core": ref["us_per_eval_1core"],
                "function_type": ref["function_type"], "measured_on": ref["measured_on"],
                "sample": f"{ref['what']}: 256-row samples of the same seeded workload, ~4 s on one core, then ~4 s with "
                          f"one reference process on each of the {ref['cores']} cores", "port": port}
    port["reference_cython"] = ref
    return port


def _free_port():
    import socket

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: start N copies of this command, one process per GPU (the contract's
    environment: RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT; HIP_VISIBLE_DEVICES is left alone - rank r opens
    device r), relay rank 0's stdout (the ONE JSON line), send every other rank's stdout to stderr.  Returns the exit
    code: 0 only if every rank exited 0; the first failing rank takes the others down (they would wait for it at the next
    barrier until the HostGroup timeout)."""
    import subprocess
    import uuid

    n = args.gpus
    if not args.dry_run:
        import abr_control_amd as a

        seen = a.device_count()
        if seen < n and not args.allow_shared_device:
            print(json.dumps({"error": f"--gpus {n} but {seen} HIP device(s) enumerated; nothing was run "
                                       f"(--allow-shared-device maps rank r to device r mod devices: a test flag, "
                                       f"not a measurement)", "devices_seen": seen, "n_gpus": n}), flush=True)
            return 1
    env0 = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                MASTER_PORT=os.environ.get("MASTER_PORT") or str(_free_port()),
                ABRK_GROUP_KEY=f"{os.getpid()}_{uuid.uuid4().hex[:12]}")
    procs = []
    for r in range(n):
        env = dict(env0, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr.fileno(), text=True))
    # rank 0's stdout is drained by a thread (a full pipe must not stall it while this loop polls)
    import threading

    out0 = []
    rd = threading.Thread(target=lambda: out0.extend(procs[0].stdout.readlines()), daemon=True)
    rd.start()
    rc, failed = 0, None
    pending = set(range(n))
    while pending:
        for r in sorted(pending):
            c = procs[r].poll()
            if c is None:
                continue
            pending.discard(r)
            if c != 0 and failed is None:
                rc, failed = (c if 0 < c < 256 else 1), r
                for o in pending:  # this rank's peers would sit in a barrier until their timeout
                    procs[o].terminate()
        time.sleep(0.02)
    rd.join(timeout=5)
    sys.stdout.write("".join(out0))
    sys.stdout.flush()
    if failed is not None:
        print(f"bench.py: the run is void - exit codes of ranks 0..{n - 1}: {[p.returncode for p in procs]} "
              f"(negative: terminated by this launcher after the first failure)", file=sys.stderr)
    return rc


def dry_run_rank(args, rank, world, group):
    """TEST leg of the self-launch (--dry-run): the ranks meet exactly as a real run's do - barrier, max of a wall time,
    gather of a per-rank record - and rank 0 prints one stub line.  No device, no kernels, nothing measured."""
    if group:
        group.barrier()
    if rank == args.dry_run_fail_rank:
        raise SystemExit(3)
    wall = 1e-3 * (1 + rank)
    seen = [{"rank": rank, "pid": os.getpid(), "local_rank": int(os.environ.get("LOCAL_RANK", 0))}]
    if group:
        wall = group.max(wall)
        seen = group.exchange(seen[0])
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "n_ranks_seen": len(seen), "ranks": seen,
                          "wall_max": wall, "steps": args.steps, "warmup": args.warmup}), flush=True)
    if group:
        group.close()


def single_process(args):
    """`python bench.py --gpus N --single-process`: ONE process, one host thread, N devices (the reference's usage
    model - one process owning the whole control loop - scaled out).  Every device holds its shard resident and its own
    recorded plan; a step is enqueued on all of them by one abrk_plans_launch (hipGraph replay of K ticks per device), then
    every stream is drained.  Same blocks as the one-process-per-GPU line: `value` (weak scaling, the workload's batch per
    device), `strong_scaling_cfg4` (global batch 2^20 cut over the devices), plus `resident_shard_step_cfg4`: the step of
    config 4 as 8 resident shards against eight plain launches of one shard's size (one-GPU boxes: all shards on device 0
    with --allow-shared-device)."""
    import abr_control_amd as a
    from abr_control_amd import engine
    from abr_control_amd.sharding import shard_range

    seen = a.device_count()
    if seen < 1:
        raise SystemExit("bench.py needs a HIP device: abr_control_amd has no CPU fallback")
    N = args.gpus
    if seen < N and not args.allow_shared_device:
        print(json.dumps({"error": f"--gpus {N} --single-process but {seen} HIP device(s) enumerated; nothing was run "
                                   f"(--allow-shared-device maps shard g to device g mod devices: a test flag, not a "
                                   f"measurement)", "devices_seen": seen, "n_gpus": N}), flush=True)
        return 1
    devices = [g % seen for g in range(N)]
    arm, B0, dts, kind, kw, _ = WORKLOADS[args.workload]
    if kind in ("rollout", "ik"):
        raise SystemExit("--single-process times recorded plans; the rollout / ik workloads have none")
    B = args.batch or B0

    def sync(runs):
        for r in runs:
            r.stream.sync()

    def timed(runs, K, warm):
        """K ticks on every device as ONE call (hipGraph of K nodes per device), wall clock around enqueue + drain"""
        plans = [r.plan for r in runs]
        if warm:
            engine.plans_launch(plans, warm, graph=False)
        engine.plans_launch(plans, K, graph=True)  # captures + instantiates every device's graph (untimed)
        sync(runs)
        t0 = time.perf_counter()
        engine.plans_launch(plans, K, graph=True)
        t_enq = time.perf_counter() - t0
        sync(runs)
        return time.perf_counter() - t0, t_enq

    runs = [Runner(args.workload, B, d, a.Stream(d)) for d in devices]
    wall, t_enq = timed(runs, args.steps, args.warmup)
    out = {
        "metric": "OSC control steps/sec (batched UR5 6-DOF)" if args.workload == "cfg2" else f"control steps/sec ({args.workload})",
        "value": round(N * B * args.steps / wall, 1), "unit": "control steps/s", "n_gpus": N, "n_devices_seen": seen,
        "devices": devices, "shared_device": len(set(devices)) < N, "process_model": "ONE process, one host thread, "
        "a resident shard + recorded plan per device, K ticks per device enqueued by one abrk_plans_launch",
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(wall / args.steps * 1e3, 6),
        "host_enqueue_us_all_devices": round(t_enq * 1e6, 1), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": dts, "data": "synthetic",
        "config": {"workload": f"{args.workload}: {arm} {kind} batch={B} per GPU, inputs resident in HBM, "
                               f"q~U(0,2pi) dq~U(0,5) target~U(-1,1) seed 1", "arm": arm, "batch_per_gpu": B,
                   "global_batch": B * N, "parallelism": f"batch-shard x{N}, no collective, single process",
                   "device": a.device_name(devices[0]), "launch": f"hipGraph replay, {args.steps} kernel nodes per device"},
    }
    del runs
    if not args.no_strong_leg:
        G = 1 << 20
        r4 = [Runner("cfg4", shard_range(G, g, N)[1] - shard_range(G, g, N)[0], d, a.Stream(d),
                     global_rows=(shard_range(G, g, N)[0], G)) for g, d in enumerate(devices)]
        k4 = max(min(args.steps, 400), 8)
        wall4, _ = timed(r4, k4, min(args.warmup, 50))
        out["strong_scaling_cfg4"] = {
            "workload": "cfg4: ur5 OSC + g + C, global batch 2^20 sharded by contiguous rows, no collective, one process",
            "global_batch": G, "n_gpus": N, "n_devices_seen": seen, "rows_per_gpu": r4[0].B, "steps": k4,
            "us_per_step": round(wall4 / k4 * 1e6, 3), "evals_per_s": round(G * k4 / wall4, 1), "scaling": "strong"}
        del r4
        # the resident 8-shard step of config 4 against eight plain launches of one shard's size on one stream
        S = 8
        rows = G // S
        rs = [Runner("cfg4", rows, devices[g % N], a.Stream(devices[g % N]), global_rows=(g * rows, G)) for g in range(S)]
        w8, enq8 = timed(rs, 200, 20)
        del rs
        one = Runner("cfg4", rows, devices[0], a.Stream(devices[0]), global_rows=(0, G))
        _, ms1 = one.timed(400, 20)
        same_dev = len(set(devices)) == 1
        out["resident_shard_step_cfg4"] = {
            "what": f"config 4's 2^20 rows as {S} resident shards of {rows} rows (one stream + one recorded plan each, "
                    f"{'all on device ' + str(devices[0]) if same_dev else 'over devices ' + str(sorted(set(devices)))}), "
                    f"200 ticks replayed by one abrk_plans_launch, wall clock per tick; beside it one shard's launch "
                    f"(HIP events, one stream)", "shards": S, "rows_per_shard": rows,
            "us_per_tick_all_shards": round(w8 / 200 * 1e6, 3), "host_enqueue_us": round(enq8 * 1e6, 1),
            "us_per_plain_launch_one_shard": round(ms1 * 1e3, 3),
            "ratio_to_eight_plain_launches": round((w8 / 200 * 1e6) / (S * ms1 * 1e3), 4) if same_dev else None}
        del one
    print(json.dumps(out), flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="rows per GPU (default: the workload's)")
    ap.add_argument("--roofline-batch", type=int, default=8 << 20, help="rows of the HBM-sized roofline leg")
    ap.add_argument("--roofline-steps", type=int, default=30)
    ap.add_argument("--sustain-seconds", type=float, default=2.0,
                    help="every HBM-sized leg then runs back to back for this long; `roofline.frac` = mean of the last "
                         "second (0: quote the short run of --roofline-steps launches, as the PMC passes do)")
    ap.add_argument("--dump-shard-u", default="", help="directory: every rank writes the sha256 of its cfg4 shard's u")
    ap.add_argument("--no-extras", action="store_true", help="skip the osc6 and shard_sweep legs of the default line")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline-leg", action="store_true")
    ap.add_argument("--no-strong-leg", action="store_true", help="skip the strong-scaling leg (cfg4, global batch 2^20)")
    ap.add_argument("--no-streams-leg", action="store_true", help="skip the concurrent_streams leg (rocprofv3's "
                    "kernel tracing crashes inside hipGraphLaunch when 16 streams replay graphs at once)")
    ap.add_argument("--allow-shared-device", action="store_true",
                    help="TEST flag: with fewer devices than ranks, rank r runs on device r mod devices (the N > 1 launch "
                         "contract on a one-GPU box) instead of the run failing with `devices_seen`")
    ap.add_argument("--dry-run", action="store_true",
                    help="TEST flag: no device is touched; the ranks only meet (barrier, max, gather) and rank 0 prints "
                         "a stub line - drives the spawn / collect / failure logic of the self-launch on a CPU box")
    ap.add_argument("--dry-run-fail-rank", type=int, default=-1, help="TEST flag: this rank exits 3 after the barrier")
    ap.add_argument("--single-process", action="store_true",
                    help="ONE process drives all --gpus N devices (resident shards, one recorded plan per device, "
                         "abrk_plans_launch) instead of one process per GPU")
    ap.add_argument("--also", default="", help="comma-separated extra workloads: one HBM-sized roofline leg each "
                                               "(same process, so one rocprofv3 session sees every kernel)")
    args = ap.parse_args()

    from abr_control_amd.sharding import dist_env

    rank, local_rank, world = dist_env()
    if args.single_process:
        if world > 1:
            raise SystemExit("--single-process under a one-process-per-GPU launcher makes no sense")
        raise SystemExit(single_process(args))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        # no launcher around this process: be the launcher (one process per GPU, this same command line)
        raise SystemExit(launch_ranks(args, sys.argv[1:]))
    # coordination only (barrier + max of the timings + one gather of per-GPU figures): a few bytes over a local socket
    # (abr_control_amd.sharding.HostGroup) - no communication library, nothing on the data path
    group = None
    if world > 1:
        from abr_control_amd.sharding import HostGroup

        # (generous: rank 0 runs its single-GPU legs - HBM-sized roofline, six-row law, shard sweep - alone while the
        #  other ranks already wait at the closing barrier)
        group = HostGroup(rank, world, timeout=1800.0)
        group.barrier()

    if args.dry_run:
        return dry_run_rank(args, rank, world, group)

    import abr_control_amd as a

    if a.device_count() < 1:
        raise SystemExit("bench.py needs a HIP device: abr_control_amd has no CPU fallback")
    if local_rank >= a.device_count() and not args.allow_shared_device:
        # never double up silently: a rank without a device of its own fails the run (the self-launcher checks before
        # it starts anything; this is the same rule under an external launcher)
        raise SystemExit(json.dumps({"error": f"rank {rank} (LOCAL_RANK {local_rank}) has no device of its own",
                                     "devices_seen": a.device_count(), "n_gpus": world}))
    device = local_rank % a.device_count()
    stream = a.Stream(device)
    arm, B0, dts, kind, kw, _ = WORKLOADS[args.workload]
    B = args.batch or B0
    run = Runner(args.workload, B, device, stream)

    barrier = group.barrier if group else None
    wall, ms = run.timed(args.steps, args.warmup, barrier)
    # the same step when the fixed cost of a (graph) launch is amortised: 2000 steps, HIP events (not `value`)
    ms_long = None
    if rank == 0 and args.steps < 1000 and kind not in ("rollout", "ik"):
        gs = run.graph_steps
        run.graph_steps = int(os.environ.get("ABRK_BENCH_GRAPH", "100"))
        _, ms_long = run.timed(2000, 0)
        run.graph_steps = gs
    ranks_seen = [{"rank": rank, "device": device, "wall_s": round(wall, 9)}]
    if group:
        # every rank's wall time for the K steps and the device it ran on: `value` takes the MAX, the list is the
        # line's own evidence of how many ranks (and which devices) took part
        ranks_seen = group.exchange(ranks_seen[0])
        wall = max(r["wall_s"] for r in ranks_seen)
    value = world * run.evals_per_launch * args.steps / wall

    out = None
    if rank == 0:
        out = {
            "metric": "OSC control steps/sec (batched UR5 6-DOF)" if args.workload == "cfg2"
            else f"control steps/sec ({args.workload})",
            "value": round(value, 1), "unit": "control steps/s", "n_gpus": world, "n_ranks_seen": len(ranks_seen),
            "ranks": ranks_seen, "shared_device": len({r["device"] for r in ranks_seen}) < len(ranks_seen),
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(wall / args.steps * 1e3, 6), "higher_is_better": True, "scaling": "weak",
            "us_per_step_long_run": None if ms_long is None else round(ms_long * 1e3, 3),
            "vs_baseline": None, "dtype": dts, "data": "synthetic",
            "config": {"workload": f"{args.workload}: {arm} {kind} batch={B} per GPU, inputs resident in HBM, "
                                   f"q~U(0,2pi) dq~U(0,5) target~U(-1,1) seed 1", "arm": arm, "batch_per_gpu": B,
                       "global_batch": B * world, "params": {k: list(v) if isinstance(v, (list, tuple)) else v for k, v in kw.items()},
                       "parallelism": f"batch-shard x{world}, no collective", "device": a.device_name(device),
                       "launch": (f"hipGraph replay, {run.graph_steps} kernel nodes (= steps) per graph launch"
                                  if getattr(run, "used_graph", False) else "one kernel launch per step")},
            "roofline_config": roofline(run, ms, f"{args.workload} batch={B} (cache-resident, launch-bound)"),
        }
    # STRONG scaling (BASELINE config 4: UR5 OSC + gravity + Coriolis, GLOBAL batch 2^20 cut into contiguous row
    # shards over the ranks): every rank evaluates its shard, same barrier + MAX-over-ranks protocol as `value`.
    # The weak-scaling `value` above keeps the per-GPU batch fixed; this leg keeps the total fixed.
    if not args.no_strong_leg:
        from abr_control_amd.sharding import shard_range

        G = 1 << 20
        lo, hi = shard_range(G, rank, world)
        r4 = Runner("cfg4", hi - lo, device, stream, global_rows=(lo, G))
        k4 = max(min(args.steps, 400), 8)
        wall4, ms4 = r4.timed(k4, min(args.warmup, 50), barrier)
        n_seen4 = 1
        if group:
            walls4 = group.exchange(float(wall4))  # every rank's wall: their count is what the line reports as seen
            wall4, n_seen4 = max(walls4), len(walls4)
        if args.dump_shard_u:  # test hook: this rank's shard of u (tests compare with the unsharded call, bit for bit)
            import hashlib

            r4.stream.sync()
            uh = r4.u.numpy()
            os.makedirs(args.dump_shard_u, exist_ok=True)
            json.dump({"rank": rank, "lo": lo, "hi": hi, "sha256": hashlib.sha256(uh.tobytes()).hexdigest(),
                       "first": uh[0].tolist(), "last": uh[-1].tolist()},
                      open(os.path.join(args.dump_shard_u, f"shard_u_rank{rank}.json"), "w"))
        if rank == 0:
            out["strong_scaling_cfg4"] = {
                "workload": "cfg4: ur5 OSC + g + C, global batch 2^20 sharded by contiguous rows, no collective",
                "global_batch": G, "n_gpus": world, "n_ranks_seen": n_seen4, "rows_per_gpu": hi - lo, "steps": k4,
                "us_per_step": round(wall4 / k4 * 1e6, 3), "evals_per_s": round(G * k4 / wall4, 1),
                "kernel_us_rank0": round(ms4 * 1e3, 3), "scaling": "strong"}
        del r4
    # HBM-sized leg per GPU (every rank; the figures are gathered on rank 0)
    if world > 1 and not args.no_roofline_leg:
        rb = args.roofline_batch
        bigr = Runner(args.workload, rb, device, stream)
        # the ranks enter the timed launches together (barrier inside timed()), and each then runs for the same
        # --sustain-seconds: the legs overlap in time, so the GPUs of the node draw power and host attention at once
        mine = roofline_leg(bigr, f"{args.workload} batch={rb} on every GPU at once (barrier-aligned)",
                            args.roofline_steps, args.sustain_seconds, barrier)
        del bigr
        gathered = group.exchange({"rank": rank, "device": device, "us_per_launch": mine["us_per_launch"],
                                   "achieved": mine["achieved"], "frac": mine["frac"]})
        if rank == 0:
            out["roofline_per_gpu"] = {"n_ranks_seen": len(gathered), "gpus": gathered}
    # HBM-sized leg for the roofline (rank 0 only; the figure is per GPU)
    if rank == 0 and not args.no_roofline_leg:
        del run
        # the iterative workloads carry [B, T, n] trajectories (ik: 19 KB per row): keep their leg at 256 k rows
        rb = min(args.roofline_batch, 1 << 18) if kind in ("ik", "rollout") else args.roofline_batch
        big = Runner(args.workload, rb, device, stream)
        # warm-up long enough to leave the boost transient behind: after idle the first ~5 launches of this kernel run
        # at boost clocks (349 us at 8 M rows), the power limiter then overshoots (540 us) and settles (~430 us) within
        # ~12 launches (rocprofv3 kernel trace, profiles/round1); the timed launches are the sustained rate
        out["roofline"] = roofline_leg(big, f"{args.workload} batch={rb} ({rb * big.bytes_per_eval / 2**20:.0f} MiB "
                                            f"algorithmic, >> 256 MiB Infinity Cache)", args.roofline_steps,
                                       args.sustain_seconds)
        del big
    elif rank == 0:
        out["roofline"] = out["roofline_config"]
    if rank == 0 and args.workload == "cfg2" and not args.no_roofline_leg:
        # the HBM-bound mode of the same path: every robot_config output of a row (Tx, J, M, g) in one launch
        full = Runner("oscF", args.roofline_batch // 2, device, stream)
        out["roofline_full_outputs"] = roofline_leg(full, f"oscF batch={full.B}: u + Tx,J,M,g per row from one launch of "
                                                          f"the fused kernel, 840 B/row", args.roofline_steps,
                                                    args.sustain_seconds)
        del full
    if rank == 0 and args.workload == "cfg2" and not args.no_roofline_leg and not args.no_extras:
        # the reference benchmark's own UR5 setting (examples/timing_plots.py:36: ctrlr_dof = [True] * 6): the six-row
        # law, config-sized step and HBM-sized leg, on the default line
        r6 = Runner("osc6", B, device, stream)
        _, ms6 = r6.timed(max(min(args.steps, 400), 8), min(args.warmup, 50))
        step6 = roofline(r6, ms6, f"osc6 batch={B} (cache-resident, launch-bound)")
        del r6
        b6 = Runner("osc6", args.roofline_batch, device, stream)
        out["osc6"] = roofline_leg(b6, f"osc6 batch={b6.B}: xyz + orientation (six task rows), the reference "
                                       f"benchmark's UR5 setting", args.roofline_steps, args.sustain_seconds)
        out["osc6"]["config_sized_step"] = {"batch": B, "us_per_step": step6["us_per_launch"],
                                            "evals_per_s": step6["evals_per_s"], "kernel": step6["kernel"]}
        del b6
        # BASELINE config 4 at the shard sizes of a 1/2/4/8-GPU node, each on THIS ONE GPU (labelled so: no scaling is
        # extrapolated): the 8-way shard of 2^20 rows is 131 072 rows = 25 MB, a launch- and cache-bound regime the
        # HBM-sized legs do not show
        sweep = []
        for parts in (1, 2, 4, 8):
            rs = Runner("cfg4", (1 << 20) // parts, device, stream)
            _, ms_s = rs.timed(200, 20)
            sweep.append({"rows": rs.B, "would_be_n_gpus": parts, "us_per_step": round(ms_s * 1e3, 3),
                          "evals_per_s": round(rs.B / (ms_s * 1e-3), 1),
                          "frac_of_hbm_peak": round(rs.B * rs.bytes_per_eval / (ms_s * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)})
            del rs
        out["shard_sweep_cfg4_single_gpu"] = {
            "what": "cfg4 (UR5 OSC + g + C) at 2^20 / {1,2,4,8} rows on ONE GPU, hipGraph replay of 200 steps, HIP-event "
                    "time per step: the per-GPU step a 1/2/4/8-GPU strong-scaling run would execute - measured "
                    "single-GPU, not a scaling result", "legs": sweep}
    if rank == 0 and args.also:
        out["also"] = {}
        for w in args.also.split(","):
            # same batch and protocol as the main leg (the iterative workloads write [B, T, n] trajectories: 256 k rows)
            rbx = min(args.roofline_batch, 1 << 18) if WORKLOADS[w][3] in ("ik", "rollout") else args.roofline_batch
            extra = Runner(w, rbx, device, stream)
            out["also"][w] = roofline_leg(extra, f"{w} batch={extra.B}", args.roofline_steps, args.sustain_seconds)
            del extra
            if w in ("osc6", "osc5_j2", "cfg3", "cfg4"):  # and the step at the workload's own (config-sized) batch
                small = Runner(w, WORKLOADS[w][1], device, stream)
                _, ms_s = small.timed(400, 50)
                st = roofline(small, ms_s, f"{w} batch={small.B} (cache-resident, launch-bound)")
                out["also"][w]["config_sized_step"] = {"batch": small.B, "us_per_step": st["us_per_launch"],
                                                       "evals_per_s": st["evals_per_s"], "kernel": st["kernel"]}
                del small
    if rank == 0 and args.workload == "cfg2" and not args.no_roofline_leg and not args.no_streams_leg:
        out["concurrent_streams"] = [concurrent_streams_rate("cfg2", B, device, s, args.steps) for s in (2, 4, 8, 16)]
    if rank == 0 and args.workload == "cfg2":
        out["parity"] = parity_vs_reference(device)
        out["host_staged"] = host_staged_rate(device)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.workload)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if group:
        group.close()


if __name__ == "__main__":
    main()
