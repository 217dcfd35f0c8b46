set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp2
(time python -m pytest tests -m gpu -x -q) > gpurun_out/exp2/pytest_gpu.log 2>&1
tail -6 gpurun_out/exp2/pytest_gpu.log
for tag in w1b64 w2b64 w3b64 w4b64 w2b256 w1b256; do
  for w in cfg2 cfg4; do
    ABRK_LIB_PATH=$GRAFT_REPO_ROOT/abr_control_amd/csrc/build/variants/libabrk_$tag.so python bench.py --workload $w --steps 500 --warmup 50 --no-cpu-baseline > gpurun_out/exp2/bench_${w}_$tag.json 2> gpurun_out/exp2/bench_${w}_$tag.err
    python - <<PY
import json
d = json.load(open("gpurun_out/exp2/bench_${w}_$tag.json"))
print("$tag $w", "cfg:", d["ms_per_step"]*1e3, "us/step", d["value"]/1e6, "Mev/s | big:", d["roofline"]["us_per_launch"], "us", d["roofline"]["evals_per_s"]/1e9, "Gev/s frac", d["roofline"]["frac"])
PY
  done
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/exp2/pmc_sq -o sq -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --roofline-steps 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/exp2/pmc_sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/exp2/pmc_grbm -o grbm -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --roofline-steps 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/exp2/pmc_grbm.log 2>&1
ls -R $GRAFT_REPO_ROOT/gpurun_out/exp2 | head -30
