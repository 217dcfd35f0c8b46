#!/bin/bash
# finish kernel: LDS asked for per workgroup as an occupancy cap (40 KiB: four wavefronts per CU)  -> gpurun_out/r4pc3/
# (ABRK_FINISH_LDS existed in the experiment build only: the cap changed nothing and was not kept - profiles/round4/NOTES.md)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4pc3; mkdir -p $O
S="--steps 400 --warmup 50 --no-roofline-leg --no-strong-leg --no-cpu-baseline --no-streams-leg --no-extras"
ab() {  # label, workload, batch, env...
  local lab=$1 w=$2 b=$3; shift 3
  env "$@" timeout 300 python bench.py --workload $w --batch $b $S 2> $O/err_$lab.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lab', '$w', 'B=$b', d['roofline_config']['us_per_launch'], 'us/step')" | tee -a $O/ab.txt
}
: > $O/ab.txt
for b in 4096 8192 16384; do
  ab ${b}_lds0 osc6 $b A=1
  ab ${b}_lds40k osc6 $b ABRK_FINISH_LDS=40960
  ab ${b}_lds80k osc6 $b ABRK_FINISH_LDS=81920
done
ab 16384_s12_lds40k osc6 16384 ABRK_FINISH_LDS=40960 ABRK_FINISH_SLOTS=12
ab 16384_s6_lds40k osc6 16384 ABRK_FINISH_LDS=40960 ABRK_FINISH_SLOTS=6
ab 32768_s8r1_lds40k osc6 32768 ABRK_FINISH_LDS=40960 ABRK_FINISH_SLOTS=8 ABRK_FINISH_ROUNDS=1
ab 32768_s8r1 osc6 32768 ABRK_FINISH_SLOTS=8 ABRK_FINISH_ROUNDS=1
ab 65536_s2r1_lds40k osc6 65536 ABRK_FINISH_LDS=40960 ABRK_FINISH_SLOTS=2 ABRK_FINISH_ROUNDS=1
ab j2_4096_lds40k osc5_j2 4096 ABRK_FINISH_LDS=40960
ab j2_4096 osc5_j2 4096 A=1
