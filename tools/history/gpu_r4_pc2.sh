#!/bin/bash
# slots per chunk / records per wavefront of the finish kernel by batch size (measurement switches)  -> gpurun_out/r4pc2/
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4pc2; mkdir -p $O
S="--steps 400 --warmup 50 --no-roofline-leg --no-strong-leg --no-cpu-baseline --no-streams-leg --no-extras"
ab() {  # label, workload, batch, env...
  local lab=$1 w=$2 b=$3; shift 3
  env "$@" timeout 300 python bench.py --workload $w --batch $b $S 2> $O/err_$lab.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lab', '$w', 'B=$b', d['roofline_config']['us_per_launch'], 'us/step')" | tee -a $O/ab.txt
}
: > $O/ab.txt
ab 8192_default osc6 8192 A=1
ab 16384_s12 osc6 16384 ABRK_FINISH_SLOTS=12
ab 32768_default osc6 32768 A=1
ab 32768_s4r1 osc6 32768 ABRK_FINISH_SLOTS=4 ABRK_FINISH_ROUNDS=1
ab 32768_s8r1 osc6 32768 ABRK_FINISH_SLOTS=8 ABRK_FINISH_ROUNDS=1
ab 32768_s6r2 osc6 32768 ABRK_FINISH_SLOTS=6 ABRK_FINISH_ROUNDS=2
ab 65536_s2r1 osc6 65536 ABRK_FINISH_SLOTS=2 ABRK_FINISH_ROUNDS=1
ab 65536_s3r1 osc6 65536 ABRK_FINISH_SLOTS=3 ABRK_FINISH_ROUNDS=1
ab 65536_s4r1 osc6 65536 ABRK_FINISH_SLOTS=4 ABRK_FINISH_ROUNDS=1
ab 131072_s2r1 osc6 131072 ABRK_HANDOVER_MAX=131072 ABRK_FINISH_SLOTS=2 ABRK_FINISH_ROUNDS=1
ab 131072_recompute osc6 131072 A=1
