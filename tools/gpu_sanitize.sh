#!/bin/bash
# Runs the host-layer stress tests of the GPU suite on the sanitizer builds of libabrk.so (csrc/Makefile: make asan tsan
# ubsan - host code only, the kernels are the product's).  On the GPU box:  bash tools/gpu_sanitize.sh [outdir]
# Logs: <outdir>/{asan,tsan,ubsan}.log + summary.txt (copied to profiles/round6/sanitizers/ when clean).
set -u
cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out/sanitizers}
mkdir -p "$OUT"
RT=/opt/rocm/lib/llvm/lib/clang/22/lib/linux
TESTS="test_gpu_concurrent_host_calls_from_threads or test_gpu_six_row_from_threads_on_default_stream or \
test_gpu_concurrent_threads_own_streams or test_gpu_plan_slots_are_recycled_and_stale_ids_rejected or \
test_gpu_six_row_many_short_lived_streams or test_gpu_sharded_call_equals_unsharded_bitwise or \
test_gpu_sharded_sliding_joint_dynamics_equal_unsharded_bitwise or test_gpu_recorded_plans_equal_direct_calls or \
test_gpu_singular_flag_is_per_stream or test_gpu_resident_shards_equal_unsharded_bitwise or \
test_gpu_resident_sharded_plan_replays_k_ticks_on_every_shard or \
test_gpu_resident_shards_from_threads_and_host_sharded_calls_on_disjoint_slots or test_gpu_merged_loops_equal_separate_loops"
: > "$OUT/summary.txt"
run() {  # name, runtime .so, environment
  local name=$1 rt=$2; shift 2
  if [ ! -f abr_control_amd/libabrk_$name.so ]; then echo "$name: libabrk_$name.so not built" | tee -a "$OUT/summary.txt"; return; fi
  case "$rt" in /*) ;; *) rt=$RT/$rt;; esac
  env "$@" LD_PRELOAD=$rt ABRK_LIB_PATH=$PWD/abr_control_amd/libabrk_$name.so \
    timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "$TESTS" > "$OUT/$name.log" 2>&1
  local rc=$?
  local reports
  reports=$(grep -c -E "ERROR: AddressSanitizer|WARNING: ThreadSanitizer|runtime error:" "$OUT/$name.log")
  echo "$name: pytest rc=$rc, sanitizer reports=$reports, $(tail -1 "$OUT/$name.log")" | tee -a "$OUT/summary.txt"
}
# ASan: the HIP runtime's own allocations are not ours to check (detect_leaks=0); ROCr maps its apertures where ASan's
# shadow gap sits (protect_shadow_gap=0)
# (gcc's runtime, not ROCm's compiler-rt: the latter intercepts HSA allocations for device-side ASan and aborts hipInit
#  on an xnack- GPU - csrc/Makefile)
GCCASAN=$(readlink -f "$(gcc -print-file-name=libasan.so)")
run asan "$GCCASAN" ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:use_sigaltstack=0
# TSan: only the instrumented module (libabrk's host layer) is judged; the interpreter and the HIP runtime are not built for it
run tsan libclang_rt.tsan-x86_64.so TSAN_OPTIONS=ignore_noninstrumented_modules=1:halt_on_error=0:report_signal_unsafe=0
run ubsan libclang_rt.ubsan_standalone-x86_64.so UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=0
cat "$OUT/summary.txt"
