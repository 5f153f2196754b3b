#!/bin/bash
# The host layer under ThreadSanitizer and AddressSanitizer on the GPU box.
#   here (no GPU):   make -C hisstools_library_amd/csrc tsan asan      -> tools/sanitize/lib/libhisstools_amd_{tsan,asan}.so
#   on the box:      bash tools/sanitize/run.sh [tsan|asan|both] [pytest args]   -> gpurun_out/sanitize/{tsan,asan}.txt + summary
# The tests are the ones that exercise the host-side concurrency: the audio-thread contract (process beside set / resize / regrow: ownership word,
# mailbox, control sections, arena), the sharded object (shard pool, per-shard engines), the fused blocks under contention (several engines and
# host threads) and the per-pair restarts.  The sanitizer runtime is preloaded into the Python process; only libhisstools_amd is instrumented.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
WHICH=${1:-both}; shift || true
TESTS=${SAN_TESTS:-"tests/test_audio_thread_contract_gpu.py tests/test_sharded_object_gpu.py::test_sharded_object_matches_unsharded_and_oracle tests/test_sharded_object_gpu.py::test_sharded_object_device_pointers tests/test_sharded_object_gpu.py::test_sharded_object_enqueue_threads tests/test_fused_block_contention_gpu.py::test_eight_concurrent_engines tests/test_fused_nxm_gpu.py::test_four_engines_at_once tests/test_pair_restart_gpu.py tests/test_gpu_parity.py::test_reference_quirks_mode_mutes_the_pair_while_set_is_in_progress"}
RTDIR=/opt/rocm/lib/llvm/lib/clang/22/lib/linux
OUT=gpurun_out/sanitize; mkdir -p $OUT
# (the sanitizers intercept dlopen, and a library opened through the interceptor is no longer looked up along its caller's RPATH: PyTorch's lazily
# opened pieces — libcaffe2_nvrtc.so at the first CUDA call — need their directory on the search path)
TORCH_LIB=$(python -c "import os, torch; print(os.path.join(os.path.dirname(torch.__file__), 'lib'))" 2>/dev/null)
export LD_LIBRARY_PATH=$TORCH_LIB:${LD_LIBRARY_PATH:-}
run() {
  san=$1; shift
  lib=$PWD/tools/sanitize/lib/libhisstools_amd_$san.so
  [ -f "$lib" ] || { echo "$lib missing: make -C hisstools_library_amd/csrc $san"; return 1; }
  if [ $san == tsan ]; then
    export TSAN_OPTIONS="suppressions=$PWD/tools/sanitize/tsan.supp exitcode=0 second_deadlock_stack=1 history_size=4 log_path=$PWD/$OUT/tsan_report"
    pre=$RTDIR/libclang_rt.tsan-x86_64.so
  else
    # (protect_shadow_gap=0: the ROCm runtime maps device memory where the tool would keep its gap; leaks: the Python process is not ours)
    export ASAN_OPTIONS="detect_leaks=0 protect_shadow_gap=0 exitcode=0 halt_on_error=0 log_path=$PWD/$OUT/asan_report"
    # (GCC's runtime, same interface version: the address sanitizer runtime ROCm's clang ships intercepts the HSA allocation calls for its own
    # device-side instrumentation and aborts under an uninstrumented HIP runtime — "out of memory" at the first hsa_amd_memory_pool_allocate)
    pre=/usr/lib/x86_64-linux-gnu/libasan.so.6
  fi
  rm -f $OUT/${san}_report.*
  LD_PRELOAD=$pre HCV_LIBRARY_PATH=$lib SAN_RUN=$san timeout ${SAN_TIMEOUT:-1500} python -m pytest $TESTS -m gpu -q -p no:cacheprovider "$@" > $OUT/$san.txt 2>&1
  echo "rc $?" >> $OUT/$san.txt
  n=$(cat $OUT/${san}_report.* 2>/dev/null | grep -c "^WARNING: ThreadSanitizer\|^==.*ERROR: AddressSanitizer")
  {
    echo "== $san: $(tail -3 $OUT/$san.txt | tr '\n' ' ')"
    echo "== $san: $n unsuppressed report(s)"
    cat $OUT/${san}_report.* 2>/dev/null | grep "^WARNING: ThreadSanitizer\|SUMMARY:\|^==.*ERROR" | sort | uniq -c | sort -rn | head -40
  } > $OUT/${san}_summary.txt
  cat $OUT/${san}_summary.txt
}
[ $WHICH == tsan ] || [ $WHICH == both ] && run tsan "$@"
[ $WHICH == asan ] || [ $WHICH == both ] && run asan "$@"
exit 0
