#!/bin/bash
# Round 4, last GPU call: the finish kernel on per-chunk packed records (wavefront (chunk, slot) asks for mask and
# record at once) against the build before it (csrc/variants/libabrk_base.so, a copy of the previous libabrk.so):
# the whole GPU suite first, then the same-box A/B and a kernel trace of the 4096-row step.   -> gpurun_out/r4pc/
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4pc; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?
tail -5 $O/pytest.log
[ $rc != 0 ] && { echo "GPU suite failed (rc=$rc): stopping here"; exit $rc; }
S="--steps 400 --warmup 50 --no-roofline-leg --no-strong-leg --no-cpu-baseline --no-streams-leg --no-extras"
OLD=$GRAFT_REPO_ROOT/abr_control_amd/csrc/variants/libabrk_base.so
ab() {  # label, workload, batch, env...
  local lab=$1 w=$2 b=$3; shift 3
  env "$@" timeout 300 python bench.py --workload $w --batch $b $S 2> $O/err_$lab.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lab', '$w', 'B=$b', d['roofline_config']['us_per_launch'], 'us/step')" | tee -a $O/ab.txt
}
: > $O/ab.txt
ab new_4096_1 osc6 4096 A=1
[ -f $OLD ] && ab old_4096_1 osc6 4096 ABRK_LIB_PATH=$OLD
ab new_4096_2 osc6 4096 A=1
for b in 16384 65536; do
  ab new_$b osc6 $b A=1
  [ -f $OLD ] && ab old_$b osc6 $b ABRK_LIB_PATH=$OLD
done
ab new_j2_4096 osc5_j2 4096 A=1
[ -f $OLD ] && ab old_j2_4096 osc5_j2 4096 ABRK_LIB_PATH=$OLD
ab new_4096_slots8 osc6 4096 ABRK_FINISH_SLOTS=8
ab new_65536_lane osc6 65536 ABRK_FINISH_ROUNDS=0
ab new_65536_r1 osc6 65536 ABRK_FINISH_ROUNDS=1
ab new_16384_lane osc6 16384 ABRK_FINISH_ROUNDS=0
cd /tmp && export TMPDIR=/tmp
ABRK_BENCH_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --workload osc6 --batch 4096 --steps 2000 --warmup 200 --no-roofline-leg --no-strong-leg --no-cpu-baseline --no-streams-leg --no-extras > $O/trace.log 2>&1
cd $GRAFT_REPO_ROOT
python - "$O" <<'PY' | tee $O/osc6_step_trace.txt
import sys, glob, pandas as pd
f = glob.glob(sys.argv[1] + "/trace/**/t_kernel_trace.csv", recursive=True)
if f:
    d = pd.read_csv(f[0]); d["us"] = (d.End_Timestamp - d.Start_Timestamp) / 1e3
    d = d[d.Kernel_Name.str.contains("double, 6|osc6_finish")]
    g = d.groupby(["Kernel_Name", "Grid_Size_X", "Grid_Size_Y", "Workgroup_Size_X"]).us
    print(pd.DataFrame(dict(n=g.size(), mean_us=g.mean(), med_us=g.median().round(2), min_us=g.min().round(2))).to_string())
PY
rm -rf $O/trace
# the bench line as the driver runs it
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-streams-leg > $O/bench_k20.json 2> $O/bench_k20.err
python -c "
import json; d=json.loads(open('$O/bench_k20.json').read().strip().splitlines()[-1]); print('K20 value', d['value'], d['ms_per_step']); o=d.get('osc6') or {}; print('osc6 8M frac', o.get('frac'), 'step', (o.get('config_sized_step') or {}).get('us_per_step'))"
