#!/bin/bash
# same-box A/B: the shipped single-wavefront workgroups vs the general form (-DABRK_OSC_MAX_WAVES=4 variant of the UR5 unit,
# run with 1 and 4 wavefronts per workgroup) at the config-sized batches   -> gpurun_out/r4/mw_ab.txt
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4; mkdir -p $O
S="--steps 2000 --warmup 200 --no-roofline-leg --no-strong-leg --no-cpu-baseline --no-streams-leg --no-extras"
V=$GRAFT_REPO_ROOT/abr_control_amd/csrc/variants/libabrk_mw4.so
: > $O/mw_ab.txt
for rep in 1 2 3; do
  for cfg in "shipped A=1" "general_w1 ABRK_LIB_PATH=$V ABRK_OSC_WAVES=1" "general_w4 ABRK_LIB_PATH=$V ABRK_OSC_WAVES=4"; do
    set -- $cfg; lab=$1; shift
    for wb in "cfg2 4096" "cfg4 4096" "cfg4 131072"; do
      set -- $wb "$@"; w=$1; b=$2; shift 2
      env "$@" timeout 300 python bench.py --workload $w --batch $b $S 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lab', '$w', 'B=$b', 'rep=$rep', d['roofline_config']['us_per_launch'], 'us/step')" | tee -a $O/mw_ab.txt
    done
  done
done
