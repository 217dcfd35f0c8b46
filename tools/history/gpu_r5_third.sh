#!/bin/bash
# Round 5, third GPU call: the whole GPU suite (NOT -x: every failure on the record), smoke, the x,y,z steps after the
# input prefetch, timeline of the new build.  -> gpurun_out/r5c/
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5c; mkdir -p $O
(time timeout 1700 python -m pytest tests -m gpu -q) > $O/pytest_gpu.log 2>&1
grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3; grep -E "^FAILED|^ERROR" $O/pytest_gpu.log | head -20
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
S="--steps 2000 --warmup 200 --no-roofline-leg --no-strong-leg --no-cpu-baseline --no-streams-leg --no-extras"
: > $O/ab.txt
ab() { local lab=$1 w=$2 b=$3; shift 3
  env "$@" timeout 300 python bench.py --workload $w --batch $b $S 2> $O/err_$lab.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lab', '$w B=$b', d['roofline_config']['us_per_launch'], 'us/step')" | tee -a $O/ab.txt
}
for rep in 1 2 3; do
  ab cfg2_4096_$rep cfg2 4096 A=1
  ab cfg4_4096_$rep cfg4 4096 A=1
  ab cfg4_131072_$rep cfg4 131072 A=1
  ab cfg3_16384_$rep cfg3 16384 A=1
done
ab osc6_4096 osc6 4096 A=1
ab osc6_16384 osc6 16384 A=1
tools/gpu_r5_timeline.sh run > $O/timeline.log 2>&1; tail -2 $O/timeline.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-streams-leg > $O/bench_k20.json 2> $O/bench_k20.err
python -c "
import json; d=json.loads(open('$O/bench_k20.json').read().strip().splitlines()[-1]); print('K20 value', d['value'], d['ms_per_step'], 'long', d['us_per_step_long_run'], 'frac', d['roofline']['frac']); o=d.get('osc6') or {}; print('osc6 8M frac', o.get('frac'), o.get('us_per_launch'), 'step', (o.get('config_sized_step') or {}).get('us_per_step')); print('sweep', [(l['rows'], l['us_per_step']) for l in d['shard_sweep_cfg4_single_gpu']['legs']])"
