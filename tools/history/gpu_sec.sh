#!/bin/bash
# GPU check of the SURVEY 8f-2 kernels: parity tests + bench lines (4096 rows and an HBM-sized batch)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/sec
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "secondary or native" 2>&1 | tail -15
for w in limits floating obstacles; do
  timeout 400 python bench.py --workload $w --steps 500 --warmup 50 --roofline-batch 4194304 --roofline-steps 10 \
    > gpurun_out/sec/bench_$w.json 2> gpurun_out/sec/bench_$w.err || tail -5 gpurun_out/sec/bench_$w.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/sec/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], 'value', d['value'], 'ms/step', d['ms_per_step'], 'cpu', d.get('cpu_baseline',{}).get('value'))
        for k in ('roofline_config','roofline'):
            r=d[k]; print('   ',k,'B',r['batch'],'us',r['us_per_launch'],'evals/s',r['evals_per_s'],'GB/s',r['achieved'],'frac',r['frac'])
    except Exception as e: print(f, 'ERR', e)
PY
