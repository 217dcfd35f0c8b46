#!/bin/bash
# Round 3, first GPU call: parity tests (incl. the new threaded six-row and two-rank bench tests), smoke, the default
# bench line (sustained protocol, osc6 + shard-sweep extras, the reference's Cython path on this box's host), and a
# short-steps run (the driver's command line).   -> gpurun_out/r3a/
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3a; mkdir -p $O
nproc > $O/host.txt; lscpu | head -20 >> $O/host.txt; cat /sys/fs/cgroup/cpu.max >> $O/host.txt 2>/dev/null
(time timeout 900 python -m pytest tests -m gpu -q -x) > $O/pytest_gpu.log 2>&1
grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
(time timeout 600 python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err) 2>&1 | grep real; tail -3 $O/bench_cfg2.err
(time timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-streams-leg > $O/bench_k20.json 2> $O/bench_k20.err) 2>&1 | grep real
python - <<'PY'
import json, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r3a"
d = json.load(open(O + "/bench_cfg2.json"))
print({k: d[k] for k in ("value", "ms_per_step", "us_per_step_long_run", "n_gpus")})
for k in ("roofline", "roofline_full_outputs", "osc6"):
    r = d[k]; print(k, r["kernel"][-40:], r["us_per_launch"], r["frac"], r.get("sustained"), r.get("short_run"))
print("osc6 step", d["osc6"]["config_sized_step"])
print("sweep", d["shard_sweep_cfg4_single_gpu"]["legs"])
print("strong", d["strong_scaling_cfg4"])
c = d["cpu_baseline"]; print("cpu", {k: v for k, v in c.items() if k not in ("sample", "port")}); print("port", {k: v for k, v in c.get("port", {}).items() if k != "sample"})
k = json.load(open(O + "/bench_k20.json")); print("K=20:", k["value"], k["ms_per_step"], k["us_per_step_long_run"], k["roofline_config"]["us_per_launch"])
PY
