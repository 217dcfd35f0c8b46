#!/bin/bash
# Round-6 evidence: parity tests, smoke, bench lines (every HBM-sized leg under the sustained protocol: back to back for
# 2 s, mean of the last second), the rocprofv3 kernel trace + stats of the SAME bench command, and separate PMC passes.
# usage (gpurun): bash tools/gpu_profiles_r6.sh   -> gpurun_out/r6/ ; then
#                 ABRK_PROFILE_COMMIT=$(git rev-parse --short HEAD) python tools/summarize_profiles.py gpurun_out/r6 profiles/round6
set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6
rm -rf $O; mkdir -p $O
nproc > $O/host.txt; lscpu | head -25 >> $O/host.txt; cat /sys/fs/cgroup/cpu.max >> $O/host.txt 2>/dev/null
(time python -m pytest tests -m gpu -q) > $O/pytest_gpu.log 2>&1
grep -E "passed|failed" $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
ALSO="--no-streams-leg --no-strong-leg --no-extras --also cfg3,cfg4,cfg5,osc6,osc5_j2,sliding_j2,oscFC"
# 1. the PMC passes (separate runs, as MI355X_MICROARCH.md prescribes), then their summary INTO profiles/round6 of this
#    copy of the tree: the bench lines below read their `roofline.traffic` / `valu` blocks from there, so every figure
#    of a committed bench JSON - time, traffic, executed instructions - is of THIS box and THESE kernels
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --roofline-steps 5 --sustain-seconds 0 --no-cpu-baseline $ALSO > $O/pmc_$c.log 2>&1
done
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/pmc_sq -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --roofline-steps 5 --sustain-seconds 0 --no-cpu-baseline $ALSO > $O/pmc_sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_grbm -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --roofline-steps 5 --sustain-seconds 0 --no-cpu-baseline $ALSO > $O/pmc_grbm.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/summarize_profiles.py $O profiles/round6 > $O/summarize_on_box.log 2>&1; tail -2 $O/summarize_on_box.log
# 2. the bench lines
python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err; tail -3 $O/bench_cfg2.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-streams-leg --no-roofline-leg > $O/bench_cfg2_k20.json 2> $O/bench_cfg2_k20.err
for w in cfg3 cfg4 cfg5 osc6 osc5_j2 sliding_j2 oscF oscFC; do python bench.py --workload $w --steps 500 --warmup 50 --no-cpu-baseline --no-strong-leg > $O/bench_$w.json 2> $O/bench_$w.err; done
ABRK_BENCH_TS=1 python bench.py --workload osc6 --steps 500 --warmup 50 --no-cpu-baseline --no-strong-leg > $O/bench_osc6_ts.json 2> $O/bench_osc6_ts.err
for w in cfg3 cfg4 cfg5; do python bench.py --workload $w --steps 500 --warmup 50 --no-strong-leg --no-roofline-leg > $O/bench_${w}_cpu.json 2> $O/bench_${w}_cpu.err; done
for w in limits floating joint obstacles rollout ik dynF dynC; do python bench.py --workload $w --steps 200 --warmup 20 --roofline-batch 4194304 --no-cpu-baseline --no-strong-leg > $O/bench_$w.json 2> $O/bench_$w.err; done
python bench.py --gpus 8 --single-process --allow-shared-device --steps 20 --warmup 5 > $O/bench_single_process_8shards.json 2> $O/bench_single_process_8shards.err
python tools/concurrent_streams_probe.py > $O/concurrent_streams.jsonl 2> $O/concurrent_streams.err
S="--steps 400 --warmup 50 --no-roofline-leg --no-strong-leg --no-cpu-baseline --no-streams-leg --no-extras"
: > $O/osc6_sizes.txt
for b in 4096 16384 65536 131072 262144 524288 1048576 2097152; do
  for mode in shipped recompute; do
    E="A=1"; [ $mode = recompute ] && E="ABRK_MEASUREMENT=1 ABRK_DENSE_MAX=0"
    [ $mode = recompute ] && [ $b -le 65536 ] && continue
    env $E python bench.py --workload osc6 --batch $b $S 2> /dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('osc6 B=$b $mode', d['roofline_config']['us_per_launch'], 'us/step')" >> $O/osc6_sizes.txt
  done
done
# 3. the kernel trace of the bench command itself
cd /tmp && export TMPDIR=/tmp
# (every leg runs its full sustained protocol, so the trace mean of a
# (kernel, grid) over ALL its launches is what the bench line's `frac` must reproduce
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline $ALSO > $O/stats.log 2>&1
# 4. the 4096-row six-row step as plain launches: first pass + finish kernel, and the round-3 scheme beside it
cd /tmp
for mode in handover round3; do
  E="A=1"; [ $mode = round3 ] && E="ABRK_MEASUREMENT=1 ABRK_NO_HANDOVER=1"
  env $E ABRK_BENCH_GRAPH=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/osc6step_$mode -o t -- python $GRAFT_REPO_ROOT/bench.py --workload osc6 --steps 200 --warmup 20 --no-roofline-leg --no-strong-leg --no-cpu-baseline --no-streams-leg --no-extras > $O/osc6step_$mode.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - "$O" > $O/osc6_step_trace.txt 2>&1 <<'PY'
import sys, glob, pandas as pd
O = sys.argv[1]
for mode in ("handover", "round3"):
    f = glob.glob(f"{O}/osc6step_{mode}/**/t_kernel_trace.csv", recursive=True)
    if not f: continue
    df = pd.read_csv(f[0]); df["kernel"] = df["Kernel_Name"].str.split("(").str[0].str.replace("void abrk::", "").str[:80]
    df["us"] = (df["End_Timestamp"] - df["Start_Timestamp"]) / 1e3
    print(mode); print(df.groupby(["kernel", "Grid_Size_X", "Workgroup_Size_X"]).agg(n=("us", "size"), mean_us=("us", "mean"), med_us=("us", "median"), min_us=("us", "min")).to_string())
PY
cat $O/osc6_step_trace.txt
# 5. soak: seeded fuzz of every kernel family against the oracle on this build (new seeds)
timeout 900 python tools/gpu_soak.py 300 130000 > $O/soak.log 2>&1; tail -4 $O/soak.log
find $O -name "*.db" -delete 2>/dev/null
# the summary of everything (kernel stats, trace means per leg, PMC means per launch, bench lines), made HERE: what comes
# back is capped at 64 MiB, and the raw counter files of the PMC passes alone exceed that
cd $GRAFT_REPO_ROOT
ABRK_PROFILE_COMMIT=${ABRK_PROFILE_COMMIT:-unknown} python tools/summarize_profiles.py $O $O/summary > $O/summarize_final.log 2>&1; tail -2 $O/summarize_final.log
find $O -path "*pmc_*" -name "*.csv" -size +512k -delete 2>/dev/null
find $O -name "*.csv" -size +12M -delete 2>/dev/null
du -sh $O; ls $O | head -80
