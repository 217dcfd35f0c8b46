#!/bin/bash
# Round 6, first GPU call: the whole GPU suite on the round-6 build (NOTS parity tests, bench-workload parity test,
# per-stream ABRK_ESINGULAR, resident shards), smoke(), the driver's bench line, the concurrent-streams probe under three
# GPU_MAX_HW_QUEUES settings.  -> gpurun_out/r6a/
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6a; mkdir -p $O
nproc > $O/host.txt; lscpu | head -20 >> $O/host.txt
(time timeout 2400 python -m pytest tests -m gpu -q -x) > $O/pytest_gpu.log 2>&1
grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
for hq in default 8 16; do
  if [ $hq = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$hq; fi
  timeout 300 python tools/concurrent_streams_probe.py > $O/streams_hwq_$hq.jsonl 2> $O/streams_hwq_$hq.err
done
unset GPU_MAX_HW_QUEUES
grep -h graph $O/streams_hwq_*.jsonl | tail -20
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_k20.json 2> $O/bench_k20.err
python -c "
import json; d=json.loads(open('$O/bench_k20.json').read().strip().splitlines()[-1]); print('K20 value', d['value'], d['ms_per_step'], 'frac', d['roofline']['frac']); o=d.get('osc6') or {}; print('osc6 8M frac', o.get('frac'), o.get('us_per_launch'), 'step', (o.get('config_sized_step') or {}).get('us_per_step')); print('sweep', [(l['rows'], l['us_per_step']) for l in d['shard_sweep_cfg4_single_gpu']['legs']])"
