#!/bin/bash
# Round 5, second GPU call: the fused single-launch prototype, the whole GPU suite, smoke, six-row steps (grouped finish
# kernel at 16384 rows against the per-chunk one), the shard sweep after the tail fix.  -> gpurun_out/r5b/
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5b; mkdir -p $O
tools/gpu_r5_fused.sh run > $O/fused.log 2>&1; tail -14 $O/fused.log
(time timeout 1500 python -m pytest tests -m gpu -q -x) > $O/pytest_gpu.log 2>&1
grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
S="--steps 400 --warmup 50 --no-roofline-leg --no-strong-leg --no-cpu-baseline --no-streams-leg --no-extras"
: > $O/ab.txt
ab() { local lab=$1 w=$2 b=$3; shift 3
  env "$@" timeout 300 python bench.py --workload $w --batch $b $S 2> $O/err_$lab.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lab', '$w B=$b', d['roofline_config']['us_per_launch'], 'us/step')" | tee -a $O/ab.txt
}
for rep in 1 2; do
  ab group16_$rep osc6 16384 A=1
  ab perchunk_$rep osc6 16384 ABRK_MEASUREMENT=1 ABRK_FINISH_GROUP=0
  ab group16_12k_$rep osc6 12288 A=1
  ab perchunk_12k_$rep osc6 12288 ABRK_MEASUREMENT=1 ABRK_FINISH_GROUP=0
done
ab group16_32k osc6 32768 ABRK_MEASUREMENT=1 ABRK_FINISH_GROUP=16
ab perchunk_32k osc6 32768 A=1
ab osc6_4096 osc6 4096 A=1
ab osc6_65536 osc6 65536 A=1
for rep in 1 2 3; do
  ab cfg4_131072_$rep cfg4 131072 A=1
  ab cfg2_131072_$rep cfg2 131072 A=1
  ab cfg2_4096_$rep cfg2 4096 A=1
done
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-streams-leg > $O/bench_k20.json 2> $O/bench_k20.err
python -c "
import json; d=json.loads(open('$O/bench_k20.json').read().strip().splitlines()[-1]); print('K20 value', d['value'], d['ms_per_step'], 'long', d['us_per_step_long_run'], 'frac', d['roofline']['frac']); o=d.get('osc6') or {}; print('osc6 8M frac', o.get('frac'), o.get('us_per_launch'), 'step', (o.get('config_sized_step') or {}).get('us_per_step')); print('sweep', [(l['rows'], l['us_per_step']) for l in d['shard_sweep_cfg4_single_gpu']['legs']])"
