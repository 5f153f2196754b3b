mkdir -p gpurun_out
{
echo "== C++ contract 128"; tests/cpp/build/audio_contract 128 1400 300
echo "== C++ contract 32"; tests/cpp/build/audio_contract 32 4200 300
echo "== C++ contract 32, control thread stalled 2 ms"; tests/cpp/build/audio_contract 32 4200 2000
} 2>&1 | tee gpurun_out/contract_cpp.log
timeout 1500 python -m pytest tests/test_audio_thread_contract_gpu.py tests/test_tail_ladder_gpu.py tests/test_fused_nxm_gpu.py -q -s 2>&1 | grep -v "^{" | tail -40 | tee gpurun_out/contract_py.log
