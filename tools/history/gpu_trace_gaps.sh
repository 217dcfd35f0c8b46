# kernel-trace of the config-sized graph replay: in-kernel duration vs gap between dependent kernels
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/gaps
mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -o tr -- python $GRAFT_REPO_ROOT/bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-roofline-leg > $O/bench.json 2> $O/err.log
python - <<'PY'
import csv, glob, os, statistics as st
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/gaps/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "osc_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
g = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(rows, rows[1:])]
p = [int(b["Start_Timestamp"]) - int(a["Start_Timestamp"]) for a, b in zip(rows, rows[1:])]
tail = slice(len(d) - 1500, None)
print("n", len(d), "dur ns med", st.median(d[tail]), "min", min(d[tail]), "| gap med", st.median(g[tail]), "min", min(g[tail]), "| period med", st.median(p[tail]))
PY
rm -rf $O/*/*.db 2>/dev/null
ls -R $O | head
