#!/bin/bash
# Round 6, fourth GPU call: the GPU suite on the build with the table sin/cos of the target angles, the EEF first pass and
# the dense finish kernel; the six-row step by batch size through the product (dense band against the recompute form);
# executed instructions per row of the Jaco2 / UR5 kernels by function; the driver's bench line.  -> gpurun_out/r6e/
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6e; mkdir -p $O
(time timeout 2400 python -m pytest tests -m gpu -q) > $O/pytest_gpu.log 2>&1
grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3
S="--steps 400 --warmup 50 --no-roofline-leg --no-strong-leg --no-cpu-baseline --no-streams-leg --no-extras"
: > $O/osc6_sizes.txt
for b in 4096 16384 65536 131072 262144 524288 1048576 2097152; do
  for mode in dense recompute; do
    E="A=1"; [ $mode = recompute ] && E="ABRK_MEASUREMENT=1 ABRK_DENSE_MAX=0"
    [ $mode = recompute ] && [ $b -le 65536 ] && continue
    env $E timeout 300 python bench.py --workload osc6 --batch $b $S 2> $O/err_${b}_$mode.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('osc6 B=$b $mode', d['roofline_config']['us_per_launch'], 'us/step')" | tee -a $O/osc6_sizes.txt
  done
done
bash tools/gpu_r6_pmc_legs.sh > $O/pmc_legs.log 2>&1; tail -30 $O/pmc_legs.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_k20.json 2> $O/bench_k20.err
python -c "
import json; d=json.loads(open('$O/bench_k20.json').read().strip().splitlines()[-1]); print('K20 value', d['value'], d['ms_per_step'], 'frac', d['roofline']['frac']); o=d.get('osc6') or {}; print('osc6 8M frac', o.get('frac'), o.get('us_per_launch'), 'step', (o.get('config_sized_step') or {}).get('us_per_step')); print('sweep', [(l['rows'], l['us_per_step']) for l in d['shard_sweep_cfg4_single_gpu']['legs']])"
