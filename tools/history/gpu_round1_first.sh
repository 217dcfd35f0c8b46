set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(time python -m pytest tests -m gpu -x -q) > gpurun_out/pytest_gpu.log 2>&1
tail -15 gpurun_out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -3 gpurun_out/smoke.log
python bench.py > gpurun_out/bench_cfg2.json 2> gpurun_out/bench_cfg2.err; tail -c 3000 gpurun_out/bench_cfg2.json; tail -5 gpurun_out/bench_cfg2.err
for w in cfg3 cfg4 cfg5; do python bench.py --workload $w --steps 300 --warmup 30 --no-cpu-baseline > gpurun_out/bench_$w.json 2> gpurun_out/bench_$w.err; tail -c 1500 gpurun_out/bench_$w.json; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_stats -o cfg2 -- python $GRAFT_REPO_ROOT/bench.py --steps 300 --warmup 30 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_fetch -o cfg2 -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --roofline-steps 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_write -o cfg2 -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --roofline-steps 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_write.log 2>&1
ls -R $GRAFT_REPO_ROOT/gpurun_out | head -40
