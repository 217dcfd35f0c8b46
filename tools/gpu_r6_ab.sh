#!/bin/bash
# Round 6: same-box A/B of header variants through tools/microbench/kernel_ab.hip (binaries built by tools/build_ab.sh in
# the build container; they travel with the snapshot).  usage: tools/gpu_r6_ab.sh <outdir> <sizes> <reps> name1 name2 ...
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/$1; sizes=$2; reps=$3; shift; shift; shift
mkdir -p $O
for rep in $(seq 1 $reps); do
  for v in "$@"; do
    timeout 300 tools/microbench/ab_$v.bin $O/${v}_$rep.json $sizes 2>> $O/log.txt
  done
done
python3 - "$O" "$@" <<'PY'
import json, sys, glob
O, names = sys.argv[1], sys.argv[2:]
rows = {}
for v in names:
    for f in sorted(glob.glob(f"{O}/{v}_*.json")):
        for L in json.load(open(f))["legs"]:
            rows.setdefault(L["rows"], {}).setdefault(v, []).append((L["us_per_step"], L["u_hash"]))
for r in sorted(rows):
    print(r, " | ".join(f"{v}: " + "/".join(f"{x[0]:.2f}" for x in rows[r].get(v, [])) + f" [{rows[r][v][0][1][-6:]}]" for v in names if v in rows[r]))
PY
