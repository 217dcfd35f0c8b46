# A/B of library builds on the same box: interleaved repetitions of the HBM-sized OSC leg
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/ab
mkdir -p $O
V=$GRAFT_REPO_ROOT/abr_control_amd/csrc/variants
for rep in 1 2 3 4; do
  for tag in "$@"; do
    ABRK_LIB_PATH=$V/libabrk_$tag.so python bench.py --workload ${ABW:-cfg2} --steps 200 --warmup 20 --roofline-steps 40 --no-cpu-baseline > $O/b_${tag}_$rep.json 2>/dev/null
  done
done
python - "$@" <<'PY'
import json, sys, statistics as st
for tag in sys.argv[1:]:
    big = [json.load(open(f"gpurun_out/ab/b_{tag}_{r}.json"))["roofline"]["us_per_launch"] for r in (1,2,3,4)]
    cfg = [json.load(open(f"gpurun_out/ab/b_{tag}_{r}.json"))["ms_per_step"]*1e3 for r in (1,2,3,4)]
    print(f"{tag:10s} big us: min {min(big):.1f} med {st.median(big):.1f} all {big} | cfg us/step: min {min(cfg):.3f} med {st.median(cfg):.3f}")
PY
