# Launch-path experiments on the config-sized step (B=4096): kernarg placement, graph length
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/lat
mkdir -p $O
run() { # tag, env...
  tag=$1; shift
  for rep in 1 2 3; do
    env "$@" python bench.py --steps 4000 --warmup 200 --no-cpu-baseline --no-roofline-leg > $O/${tag}_$rep.json 2>$O/${tag}_$rep.err
  done
  python - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
v = [json.load(open(f"gpurun_out/lat/{tag}_{r}.json"))["ms_per_step"] * 1e3 for r in (1, 2, 3)]
k = [json.load(open(f"gpurun_out/lat/{tag}_{r}.json"))["roofline_config"]["us_per_launch"] for r in (1, 2, 3)]
print(f"{tag:28s} wall us/step {['%.3f' % x for x in v]}  event us/step {['%.3f' % x for x in k]}")
PY
}
run base            X=1
run devkernarg1     HIP_FORCE_DEV_KERNARG=1
run devkernarg0     HIP_FORCE_DEV_KERNARG=0
run graph500        ABRK_BENCH_GRAPH=500
run graph2000       ABRK_BENCH_GRAPH=2000
run eager           ABRK_BENCH_GRAPH=0
run eager_dk1       ABRK_BENCH_GRAPH=0 HIP_FORCE_DEV_KERNARG=1
run eager_dk0       ABRK_BENCH_GRAPH=0 HIP_FORCE_DEV_KERNARG=0
