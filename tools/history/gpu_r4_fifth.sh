#!/bin/bash
# Round 4, fifth GPU call: AvoidObstacles with the heavy pairs redistributed through LDS vs the one-pass kernel; the GPU
# parity suite on the final kernels.   -> gpurun_out/r4f/
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4f; mkdir -p $O
: > $O/ab_obs.txt
for rep in 1 2; do for e in A=1 ABRK_OBS_PLAIN=1; do
  env $e timeout 300 python bench.py --workload obstacles --steps 200 --warmup 20 --roofline-batch 4194304 --no-cpu-baseline --no-strong-leg 2>> $O/err.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$e rep=$rep', 'step4096', d['roofline_config']['us_per_launch'], '4M', r['us_per_launch'], r['frac'], r['kernel'][:24])" | tee -a $O/ab_obs.txt
done; done
(time timeout 1500 python -m pytest tests -m gpu -q -x) > $O/pytest_gpu.log 2>&1
grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3
