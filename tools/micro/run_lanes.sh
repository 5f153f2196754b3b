mkdir -p gpurun_out
for w in c4 ns64 c5; do
  for lanes in 0 1 0 1; do
    echo "== $w lanes $lanes: $(HCV_STREAM_LANES=$lanes python bench.py --workload $w --steps 40 --warmup 5 --also '' --no-cpu-baseline --no-all-cores --batched-block 0 --extended-ratio 0 --realtime-block 0 --offline-hops 0 --no-self-check 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], 'mac', r.get('avg_launch_ms'), 'frac', r.get('frac'), 'whole', r.get('whole_step_frac'))")"
  done
done 2>&1 | tee gpurun_out/lanes_ab.log
timeout 900 python -m pytest tests/test_steady_state_gpu.py tests/test_spectral_ir.py tests/test_cpp_dropin.py tests/test_order_check_gpu.py -m gpu -q -x 2>&1 | tail -6 | tee gpurun_out/lanes_tests.log
