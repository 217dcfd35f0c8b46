#!/bin/bash
# same-box A/B of two builds of libabrk.so (ABRK_LIB_PATH): abr_control_amd/csrc/variants/libabrk_prev.so (the previous
# commit's kernels, built by hand) against the tree's library.  usage: gpu_ab_lib.sh "<workloads>" [rounds]
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/ablib; mkdir -p $O
PREV=$GRAFT_REPO_ROOT/abr_control_amd/csrc/variants/libabrk_prev.so
for round in $(seq 1 ${2:-2}); do
  for w in $1; do
    for lib in prev new; do
      if [ $lib = prev ]; then export ABRK_LIB_PATH=$PREV; else unset ABRK_LIB_PATH; fi
      python bench.py --workload $w --steps 300 --warmup 50 --no-cpu-baseline --no-strong-leg --no-streams-leg --no-extras --sustain-seconds 1.5 > $O/${w}_${lib}_$round.json 2>$O/${w}_${lib}_$round.err
      python - $O/${w}_${lib}_$round.json $w $lib <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
print(sys.argv[2], sys.argv[3], "step us", round(d["ms_per_step"]*1e3,3), "| 8M:", r["us_per_launch"], "us frac", round(r["frac"],4))
PY
    done
  done
done
