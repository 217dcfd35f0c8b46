# Collects the round's committed evidence: parity tests, smoke, bench lines, rocprofv3 kernel stats and PMC passes.
set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r1
mkdir -p $O
(time python -m pytest tests -m gpu -q) > $O/pytest_gpu.log 2>&1
grep -E "passed|failed" $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err; tail -3 $O/bench_cfg2.err
for w in cfg3 cfg4 cfg5; do python bench.py --workload $w --steps 500 --warmup 50 --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err; done
for w in limits floating obstacles rollout ik dynF; do python bench.py --workload $w --steps 200 --warmup 20 --roofline-batch 4194304 --roofline-steps 10 > $O/bench_$w.json 2> $O/bench_$w.err; done
ALSO="--no-streams-leg --also limits,floating,obstacles,cfg3,cfg4,cfg5"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline $ALSO > $O/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --roofline-steps 5 --no-cpu-baseline $ALSO > $O/pmc_$c.log 2>&1
done
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/pmc_sq -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --roofline-steps 5 --no-cpu-baseline $ALSO > $O/pmc_sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_grbm -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --roofline-steps 5 --no-cpu-baseline $ALSO > $O/pmc_grbm.log 2>&1
ls -R $O | head -50
