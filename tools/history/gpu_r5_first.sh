#!/bin/bash
# Round 5, first GPU call: the whole GPU suite on the round-5 build (shared truncating-pinv routine, NOTS in every pass,
# ABRK_ESINGULAR, self-launching bench), smoke(), the per-wavefront timeline of the shard-sized step
# (tools/microbench/shard_step_timeline), the six-row steps, the bench line as the driver runs it.  -> gpurun_out/r5a/
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5a; mkdir -p $O
nproc > $O/host.txt; lscpu | head -20 >> $O/host.txt
tools/gpu_r5_timeline.sh run > $O/timeline.log 2>&1; tail -2 $O/timeline.log
(time timeout 1500 python -m pytest tests -m gpu -q -x) > $O/pytest_gpu.log 2>&1
grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
S="--steps 400 --warmup 50 --no-roofline-leg --no-strong-leg --no-cpu-baseline --no-streams-leg --no-extras"
: > $O/ab.txt
for b in 4096 16384 65536; do
  timeout 300 python bench.py --workload osc6 --batch $b $S 2> $O/err_$b.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('osc6 B=$b', d['roofline_config']['us_per_launch'], 'us/step')" | tee -a $O/ab.txt
done
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-streams-leg > $O/bench_k20.json 2> $O/bench_k20.err
python -c "
import json; d=json.loads(open('$O/bench_k20.json').read().strip().splitlines()[-1]); print('K20 value', d['value'], d['ms_per_step'], 'long', d['us_per_step_long_run'], 'frac', d['roofline']['frac']); o=d.get('osc6') or {}; print('osc6 8M frac', o.get('frac'), o.get('us_per_launch'), 'step', (o.get('config_sized_step') or {}).get('us_per_step')); print('sweep', [(l['rows'], l['us_per_step']) for l in d['shard_sweep_cfg4_single_gpu']['legs']])"
timeout 120 python bench.py --gpus 2 --steps 20 --warmup 5; echo "plain --gpus 2 rc=$?"
