#!/bin/bash
# same-box A/B of the x,y,z kernel (timeline tool, light stamps): A = before the tail fix (cyclic Jacobi on the truncating
# rows), B = sym3_eig, C = B + inputs requested ahead of the table fill.  -> gpurun_out/r5_tail_ab/
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_tail_ab; mkdir -p $O
for rep in 1 2 3; do for v in A B C; do timeout 120 tools/microbench/tl_$v.bin $O/${v}_$rep.json 100; done; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5_tail_ab/*.json")):
    d = json.load(open(f))
    print(f.split("/")[-1], " ".join(f"{L['rows']}/C{L['use_C']}: period {L['first_entry_to_next_first_entry_us']['median']:.2f} span {L['node_span_first_entry_to_last_exit_us']['median']:.2f} life p90 {L['wavefront_lifetime_us']['p90']:.2f} max {L['wavefront_lifetime_us']['max']:.2f} ev {L['hip_event_us_per_node']:.2f} |" for L in d["legs"]))
PY
