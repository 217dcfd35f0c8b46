set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp3
for tag in p1024 p2048 p4096 p16384 pinf; do
  for w in cfg2 cfg4; do
    ABRK_LIB_PATH=$GRAFT_REPO_ROOT/abr_control_amd/csrc/build/variants/libabrk_$tag.so python bench.py --workload $w --steps 500 --warmup 50 --no-cpu-baseline > gpurun_out/exp3/bench_${w}_$tag.json 2> gpurun_out/exp3/bench_${w}_$tag.err
    python - <<PY
import json
d = json.load(open("gpurun_out/exp3/bench_${w}_$tag.json"))
print("$tag $w", "cfg:", d["ms_per_step"]*1e3, "us/step", d["value"]/1e6, "Mev/s | big:", d["roofline"]["us_per_launch"], "us", d["roofline"]["evals_per_s"]/1e9, "Gev/s frac", d["roofline"]["frac"])
PY
  done
done
ABRK_LIB_PATH=$GRAFT_REPO_ROOT/abr_control_amd/csrc/build/variants/libabrk_p1024.so python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -5
cd /tmp && export TMPDIR=/tmp
ABRK_LIB_PATH=$GRAFT_REPO_ROOT/abr_control_amd/csrc/build/variants/libabrk_p1024.so rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/exp3/pmc_sq -o sq -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --roofline-steps 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/exp3/pmc_sq.log 2>&1
