#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4pf; mkdir -p $O
: > $O/ab.txt
for rep in 1 2 3; do for v in base pf; do
  L=$GRAFT_REPO_ROOT/abr_control_amd/libabrk.so; [ $v != base ] && L=$GRAFT_REPO_ROOT/abr_control_amd/csrc/variants/libabrk_$v.so
  ABRK_LIB_PATH=$L timeout 300 python bench.py --workload osc6 --steps 1000 --warmup 100 --no-strong-leg --no-cpu-baseline --no-streams-leg --no-extras 2>> $O/err.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v rep=$rep', 'step4096', d['roofline_config']['us_per_launch'], '8M', r['us_per_launch'], r['frac'])" | tee -a $O/ab.txt
done; done
