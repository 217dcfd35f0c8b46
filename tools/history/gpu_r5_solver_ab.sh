#!/bin/bash
# same-box A/B of the six-row law's eigen-solver: abr_control_amd/libabrk_prev.so (the commit before: full QL in every
# form) against the shipped library (early-exit QL + tridiagonal solve).  -> gpurun_out/r5_solver_ab/ab.txt
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_solver_ab; mkdir -p $O; : > $O/ab.txt
S="--steps 400 --warmup 50 --no-roofline-leg --no-strong-leg --no-cpu-baseline --no-streams-leg --no-extras"
one() { local lab=$1 w=$2 b=$3; shift 3
  env "$@" timeout 300 python bench.py --workload $w --batch $b $S 2>> $O/err.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lab', '$w B=$b', d['roofline_config']['us_per_launch'], 'us/step')" | tee -a $O/ab.txt
}
PREV=ABRK_LIB_PATH=$GRAFT_REPO_ROOT/abr_control_amd/libabrk_prev.so
for rep in 1 2; do
  for b in 4096 16384 32768 65536; do
    one new_$rep osc6 $b A=1
    one prev_$rep osc6 $b $PREV
  done
  for b in 4096 16384 65536; do
    one new_$rep osc5_j2 $b A=1
    one prev_$rep osc5_j2 $b $PREV
  done
done
for lib in new prev; do
  E=A=1; [ $lib = prev ] && E=$PREV
  env $E timeout 300 python bench.py --workload osc6 --steps 200 --warmup 20 --no-strong-leg --no-cpu-baseline --no-streams-leg --no-extras 2>> $O/err.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$lib osc6 8M', r['us_per_launch'], r['frac'])" | tee -a $O/ab.txt
  env $E timeout 300 python bench.py --workload osc5_j2 --steps 200 --warmup 20 --no-strong-leg --no-cpu-baseline --no-streams-leg --no-extras 2>> $O/err.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$lib osc5_j2 8M', r['us_per_launch'], r['frac'])" | tee -a $O/ab.txt
done
