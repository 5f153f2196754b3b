mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_offline_mfma_gpu.py -x -q 2>&1 | tail -15 | tee gpurun_out/mfma_tests.log
timeout 900 python bench.py --steps 20 --warmup 5 --also ns64,c4 > gpurun_out/bench_offline.json 2> gpurun_out/bench_offline.err
tail -c 3000 gpurun_out/bench_offline.json
cp gpurun_out/bench_details.json gpurun_out/bench_offline_details.json
tail -5 gpurun_out/bench_offline.err
