#!/bin/bash
# A/B of library variants on one box, HBM-sized sustained legs: usage  tools/gpu_ab_big.sh "<workloads>" tag1 tag2 ...
# (tag "base" = the product library; others = abr_control_amd/csrc/variants/libabrk_<tag>.so)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/ab3; mkdir -p $O
W="$1"; shift
V=$GRAFT_REPO_ROOT/abr_control_amd/csrc/variants
for rep in 1 2; do for w in $W; do for tag in "$@"; do
  L=$V/libabrk_$tag.so; [ $tag = base ] && L=$GRAFT_REPO_ROOT/abr_control_amd/libabrk.so
  ABRK_LIB_PATH=$L python bench.py --workload $w --steps 400 --warmup 40 --no-cpu-baseline --no-strong-leg --no-streams-leg --no-extras --sustain-seconds 1.5 > $O/${w}_${tag}_$rep.json 2>/dev/null
done; done; done
python - "$W" "$@" <<'PY'
import json, sys
ws, tags = sys.argv[1].split(), sys.argv[2:]
for w in ws:
    for tag in tags:
        out = []
        for rep in (1, 2):
            try:
                d = json.load(open(f"gpurun_out/ab3/{w}_{tag}_{rep}.json")); r = d["roofline"]
                out.append(f"step {d['ms_per_step']*1e3:.3f} big {r['us_per_launch']:.1f} ({r['frac']:.3f})")
            except Exception as e:
                out.append(f"failed {e}")
        print(f"{w:8s} {tag:10s} " + " | ".join(out))
PY
