#!/bin/bash
# Round 6, sixth GPU call: the GPU suite on the build where every pass of an end-effector launch takes the EEF form; bench line.
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6g; mkdir -p $O
(time timeout 2400 python -m pytest tests -m gpu -q) > $O/pytest_gpu.log 2>&1
grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_k20.json 2> $O/bench_k20.err
python -c "
import json; d=json.loads(open('$O/bench_k20.json').read().strip().splitlines()[-1]); print('K20 value', d['value'], d['ms_per_step'], 'frac', d['roofline']['frac']); o=d.get('osc6') or {}; print('osc6 8M frac', o.get('frac'), o.get('us_per_launch'), 'step', (o.get('config_sized_step') or {}).get('us_per_step')); print('sweep', [(l['rows'], l['us_per_step']) for l in d['shard_sweep_cfg4_single_gpu']['legs']])"
