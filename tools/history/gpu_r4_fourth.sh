#!/bin/bash
# Round 4, fourth GPU call: rows of Y held at a time in the six-row law (3 = shipped, 6, 2) at 8 M rows and 4096 rows;
# why the ASan build dies at start on the GPU box.   -> gpurun_out/r4e/
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4e; mkdir -p $O
: > $O/ab_yb.txt
for rep in 1 2; do for v in base yb6 yb2; do
  L=$GRAFT_REPO_ROOT/abr_control_amd/libabrk.so; [ $v != base ] && L=$GRAFT_REPO_ROOT/abr_control_amd/csrc/variants/libabrk_$v.so
  ABRK_LIB_PATH=$L timeout 300 python bench.py --workload osc6 --steps 400 --warmup 50 --no-strong-leg --no-cpu-baseline --no-streams-leg --no-extras 2>> $O/err.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v rep=$rep', 'step4096', d['roofline_config']['us_per_launch'], '8M', r['us_per_launch'], r['frac'])" | tee -a $O/ab_yb.txt
done; done
RT=/opt/rocm/lib/llvm/lib/clang/22/lib/linux
ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:verbosity=1 LD_PRELOAD=$RT/libclang_rt.asan-x86_64.so ABRK_LIB_PATH=$PWD/abr_control_amd/libabrk_asan.so \
  timeout 120 python -c "
import abr_control_amd as a
print('devices', a.device_count(), flush=True)
print(a.device_name(0), flush=True)
s = a.Stream(0); print('stream ok', flush=True)
" > $O/asan_probe.log 2>&1; echo "asan probe rc=$?"; tail -15 $O/asan_probe.log
