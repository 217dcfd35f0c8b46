#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4h; mkdir -p $O
G=$(readlink -f "$(gcc -print-file-name=libasan.so)")
TESTS="test_gpu_concurrent_host_calls_from_threads or test_gpu_six_row_from_threads_on_default_stream or test_gpu_concurrent_threads_own_streams or test_gpu_plan_slots_are_recycled_and_stale_ids_rejected or test_gpu_six_row_many_short_lived_streams or test_gpu_sharded_call_equals_unsharded_bitwise or test_gpu_sharded_sliding_joint_dynamics_equal_unsharded_bitwise or test_gpu_recorded_plans_equal_direct_calls"
ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:use_sigaltstack=0:log_path=$O/asan_report LD_PRELOAD=$G ABRK_LIB_PATH=$PWD/abr_control_amd/libabrk_asan.so \
  timeout 600 python -m pytest tests/test_gpu_parity.py -v -x -s -p no:cacheprovider -k "$TESTS" > $O/asan_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/asan_pytest.log; head -60 $O/asan_report*
