#!/bin/bash
# A/B of the six-row law on one box: product build (Jacobian rows in LDS, first pass at two waves per SIMD) against
# km6w1 (LDS rows, one wave per SIMD) and km6reg (rows in registers - the round-2 register form of the law)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3ab; mkdir -p $O
V=$GRAFT_REPO_ROOT/abr_control_amd/csrc/variants
for rep in 1 2; do
  for tag in base km6w1 km6reg; do
    L=$V/libabrk_$tag.so; [ $tag = base ] && L=$GRAFT_REPO_ROOT/abr_control_amd/libabrk.so
    ABRK_LIB_PATH=$L timeout 300 python bench.py --workload osc6 --steps 400 --warmup 40 --no-cpu-baseline --no-strong-leg --no-streams-leg --sustain-seconds 1.5 > $O/osc6_${tag}_$rep.json 2> $O/osc6_${tag}_$rep.err
  done
done
python - <<'PY'
import json, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r3ab"
for tag in ("base", "km6w1", "km6reg"):
    for rep in (1, 2):
        try:
            d = json.load(open(f"{O}/osc6_{tag}_{rep}.json"))
        except Exception as e:
            print(tag, rep, "failed", e); continue
        r = d["roofline"]
        print(f"{tag:8s} rep {rep}: step(4096) {d['ms_per_step']*1e3:.2f} us | 8M rows sustained {r['us_per_launch']:.1f} us frac {r['frac']:.3f} short {r['short_run']['us_per_launch']:.1f}")
PY
