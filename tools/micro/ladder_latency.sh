# Round 6: the paced real-time behaviour of the 64 x 64 / 10 s shape on its ladder at ratios 8 / 4 / 2 — the one measurement the
# automatic rule was waiting for.  128- and 32-sample synchronous calls through 48 tail hops (every rung's boundary several times).
mkdir -p gpurun_out
for r in 8 4 2; do
  for b in 128 32; do
    echo "ratio $r: $(TAIL_RATIO=$r SLOW_MS=0.5 timeout 600 python tools/latency.py ns64 $b 48 2>&1 | grep -v amdgpu.ids | tr '\n' ' ' | cut -c1-900)"
  done
done 2>&1 | tee gpurun_out/ladder_latency.log
