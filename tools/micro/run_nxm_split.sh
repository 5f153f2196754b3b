mkdir -p gpurun_out
{
for w in c4s8 c4g; do
  for sp in 0 1 0 1; do
    echo "== $w split $sp: $(HCV_NXM_SPLIT=$sp python bench.py --workload $w --steps 200 --warmup 20 --also '' --no-cpu-baseline --no-all-cores --batched-block 0 --extended-ratio 0 --realtime-block 0 --offline-hops 0 --no-self-check 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], 'mac', r.get('avg_launch_ms'), 'frac', r.get('frac'))")"
  done
done
timeout 1500 python -m pytest tests/test_fused_nxm_gpu.py tests/test_order_check_gpu.py tests/test_sharded_object_gpu.py tests/test_fused_block_contention_gpu.py -m gpu -q -x 2>&1 | grep -v "^{" | tail -6
PMC_COMMIT=$PMC_COMMIT PMC_WORKLOADS="" bash tools/r06_pmc.sh 2>&1 | tail -4
cp gpurun_out/pmc_json/traffic_c5_offline.json gpurun_out/traffic_c5_offline_fixed.json
} 2>&1 | tee gpurun_out/nxm_split.log
