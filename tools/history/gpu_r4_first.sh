#!/bin/bash
# Round 4, first GPU call: the six-row law in its hand-over form (first pass + wave-cooperative finish kernel) against the
# round-3 scheme on the same box, the GPU parity suite (fp32 kernels vs the reference, hand-over near-singular postures,
# finish-form bit equality, stream lifetime), the default bench line.   -> gpurun_out/r4a/
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4a; mkdir -p $O
nproc > $O/host.txt; lscpu | head -20 >> $O/host.txt
S="--steps 400 --warmup 50 --no-roofline-leg --no-strong-leg --no-cpu-baseline --no-streams-leg --no-extras"
ab() {  # label, batch, env...
  local lab=$1 b=$2; shift 2
  env "$@" timeout 300 python bench.py --workload osc6 --batch $b $S 2> $O/ab_$lab.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lab', 'B=$b', d['roofline_config']['us_per_launch'], 'us/step', d['roofline_config']['kernel'][-30:])" | tee -a $O/ab_osc6.txt
}
: > $O/ab_osc6.txt
for b in 4096 16384 65536; do
  ab handover_$b $b A=1
  ab round3_$b $b ABRK_NO_HANDOVER=1
  ab lane_$b $b ABRK_FINISH_COOP_MAX=0
done
ab grid128_4096 4096 ABRK_FINISH_GRID=128
ab coop8k_65536 65536 ABRK_FINISH_COOP_MAX=8192 ABRK_FINISH_GRID=1024
ab handover_262144 262144 A=1
ab round3_262144 262144 ABRK_NO_HANDOVER=1
# Jaco2 five rows (timing_plots.py:37)
for e in A=1 ABRK_NO_HANDOVER=1; do
  env $e timeout 300 python bench.py --workload osc5_j2 $S 2>> $O/ab_j2.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('osc5_j2 $e', d['roofline_config']['us_per_launch'], 'us/step')" | tee -a $O/ab_osc6.txt
done
(time timeout 1500 python -m pytest tests -m gpu -q -x) > $O/pytest_gpu.log 2>&1
grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
(time timeout 900 python bench.py --also osc6,osc5_j2 > $O/bench_cfg2.json 2> $O/bench_cfg2.err) 2>&1 | grep real; tail -3 $O/bench_cfg2.err
python - <<'PY'
import json, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r4a"
d = json.load(open(O + "/bench_cfg2.json"))
print({k: d[k] for k in ("value", "ms_per_step", "us_per_step_long_run", "n_gpus")})
for k in ("roofline", "roofline_full_outputs", "osc6"):
    r = d[k]; print(k, r["kernel"][-40:], r["us_per_launch"], r["frac"])
print("osc6 step", d["osc6"]["config_sized_step"])
for k, r in d["also"].items(): print("also", k, r["us_per_launch"], r["frac"], r.get("config_sized_step"))
print("sweep", d["shard_sweep_cfg4_single_gpu"]["legs"])
c = d["cpu_baseline"]; print("cpu", {k: v for k, v in c.items() if k not in ("sample", "port")})
PY
