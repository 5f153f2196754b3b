# Round 6, VERDICT r5 item 6.  (1) the runtime-only programme, every mode, under glibc's heap checks; (2) the library with stream destruction
# restored (HCV_STREAM_POOL=0) under the same checks: the create / destroy cycles of tests/test_gpu_parity.py with the helping path forced, N runs.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
export MALLOC_CHECK_=3 MALLOC_PERTURB_=165
for m in 0 1 2; do timeout 600 tools/micro/build/stream_destroy_repro $m ${CYCLES:-3000}; echo "   rc $?"; done
N=${N:-5} bash tools/micro/crash_loop.sh "HCV_STREAM_POOL=0" "HCV_STREAM_POOL=1"
} 2>&1 | tee gpurun_out/stream_destroy.log
