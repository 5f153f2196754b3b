mkdir -p gpurun_out/final
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/final/gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.txt 2>&1
python bench.py --steps 20 --warmup 5 2>/dev/null | grep '^{' > gpurun_out/final/r06_c5_bench_default_head.json
cp gpurun_out/bench_details.json gpurun_out/final/r06_c5_bench_default_head_details.json
PMC_COMMIT=$PMC_COMMIT bash tools/r06_pmc.sh > gpurun_out/final/pmc.log 2>&1
cp gpurun_out/pmc_json/traffic_c5.json gpurun_out/pmc_json/traffic_ns64.json gpurun_out/pmc_json/traffic_c4.json gpurun_out/pmc_json/traffic_c4s8.json gpurun_out/pmc_json/traffic_c5_offline.json gpurun_out/final/
tail -3 gpurun_out/final/gpu_suite.txt; cat gpurun_out/final/smoke.txt | tail -1; tail -8 gpurun_out/final/pmc.log
