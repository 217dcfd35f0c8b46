# kernel trace of the six-row OSC law at an HBM-sized batch: durations of the first pass (grid = rows / 64) and of the
# deferred second pass (grid = 2048 blocks) by launch
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/osc6split
mkdir -p $O; rm -rf $O/trace
cd /tmp && export TMPDIR=/tmp
for tag in ${TAGS:-frob trace}; do
ABRK_LIB_PATH=$GRAFT_REPO_ROOT/abr_control_amd/csrc/variants/libabrk_$tag.so rocprofv3 --kernel-trace --output-format csv -d $O/trace_$tag -o t -- python $GRAFT_REPO_ROOT/bench.py --workload osc6 --steps 20 --warmup 5 --roofline-steps 12 --no-cpu-baseline --no-streams-leg --no-strong-leg > $O/log_$tag.txt 2>&1
python - $O/trace_$tag $tag <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
by = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "double, 6" not in r["Kernel_Name"]: continue
    g = int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r["Grid_Size"])
    by[g].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for g, v in sorted(by.items()):
    v2 = v[len(v)//2:]
    print(sys.argv[2], "grid", g, "n", len(v), "mean of last half %.1f us" % (sum(v2)/len(v2)))
PY
done
