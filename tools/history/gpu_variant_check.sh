# run the GPU tests and two bench workloads against a variant library (ABRK_LIB_PATH); the "native library" test
# is skipped because it looks for the product's file name
cd $GRAFT_REPO_ROOT
export ABRK_LIB_PATH=$GRAFT_REPO_ROOT/abr_control_amd/csrc/variants/libabrk_$1.so
timeout 400 python -m pytest tests -m gpu -x -q -k "not native_library" 2>&1 | tail -8
for w in osc6 obstacles; do timeout 200 python bench.py --workload $w --no-cpu-baseline --no-strong-leg --no-streams-leg 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['config']['workload'][:30], d['ms_per_step'], r['us_per_launch'], r['evals_per_s'], r['batch'])"; done
