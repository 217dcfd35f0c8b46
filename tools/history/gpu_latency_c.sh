#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3lat; mkdir -p $O
run() { tag=$1; shift
  for rep in 1 2 3; do env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline-leg --no-strong-leg --no-extras --no-streams-leg > $O/${tag}_$rep.json 2>/dev/null; done
  python - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
d = [json.load(open(f"gpurun_out/r3lat/{tag}_{r}.json")) for r in (1, 2, 3)]
print(f"{tag:22s} wall us/step {[round(x['ms_per_step']*1e3, 3) for x in d]}  events us/step {[x['roofline_config']['us_per_launch'] for x in d]}  long-run {[x['us_per_step_long_run'] for x in d]}")
PY
}
run blocking_sync X=1
run repeat_c ABRK_BENCH_REPEAT_C=1
run repeat_c_spin ABRK_BENCH_REPEAT_C=1 ABRK_SYNC_SPIN=1
