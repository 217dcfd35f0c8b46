#!/usr/bin/env python3
"""VERDICT r5 "Next" #7: why do independent control loops stop overlapping beyond two streams?

  python tools/concurrent_streams_probe.py [--streams 1,2,4,8,16] [--steps 2000]

For each stream count: N independent config-sized batches (BASELINE config 2: UR5, 4096 rows, fp64), each with its own
stream, buffers and recorded plan (bench.Runner), driven from ONE host thread
  graph : 100-node hipGraph replays, round-robin over the streams (bench.py's `concurrent_streams` leg)
  plain : one abrk_plan_launch per step and stream, round-robin
  merged: the same N x 4096 rows as ONE launch on one stream (what engine-level merging of the loops would run)
One JSON line per (mode, N).  Run it under different GPU_MAX_HW_QUEUES values (the HIP runtime maps streams onto that many
hardware queues, default 4; read at runtime initialisation, so one process per value): tools/gpu_r6_streams.sh."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", default="1,2,4,8,16")
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--workload", default="cfg2")
    args = ap.parse_args()
    import abr_control_amd as a

    hwq = os.environ.get("GPU_MAX_HW_QUEUES", "default")
    for n in [int(x) for x in args.streams.split(",")]:
        streams = [a.Stream(0) for _ in range(n)]
        runs = [bench.Runner(args.workload, args.batch, 0, st) for st in streams]
        G = 100
        reps = max(args.steps // G, 1)
        for r in runs:
            r.plan.launch_graph(G)
        for st in streams:
            st.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            for r in runs:
                r.plan.launch_graph(G)
        for st in streams:
            st.sync()
        w = time.perf_counter() - t0
        steps = reps * G
        print(json.dumps({"mode": "graph", "hw_queues": hwq, "streams": n, "us_per_step_per_stream": round(w / steps * 1e6, 3),
                          "G_evals_per_s": round(n * args.batch * steps / w / 1e9, 3)}), flush=True)
        for r in runs:
            r.plan.launch()
        for st in streams:
            st.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            for r in runs:
                r.plan.launch()
        t_enq = time.perf_counter() - t0
        for st in streams:
            st.sync()
        w = time.perf_counter() - t0
        print(json.dumps({"mode": "plain", "hw_queues": hwq, "streams": n, "us_per_step_per_stream": round(w / steps * 1e6, 3),
                          "host_enqueue_us_per_launch": round(t_enq / (steps * n) * 1e6, 3),
                          "G_evals_per_s": round(n * args.batch * steps / w / 1e9, 3)}), flush=True)
        del runs, streams
        st = a.Stream(0)
        m = bench.Runner(args.workload, args.batch * n, 0, st)
        m.plan.launch_graph(G)
        st.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            m.plan.launch_graph(G)
        st.sync()
        w = time.perf_counter() - t0
        print(json.dumps({"mode": "merged", "hw_queues": hwq, "streams": n, "rows": args.batch * n,
                          "us_per_step": round(w / steps * 1e6, 3),
                          "G_evals_per_s": round(n * args.batch * steps / w / 1e9, 3)}), flush=True)
        del m, st


if __name__ == "__main__":
    main()
