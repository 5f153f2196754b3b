mkdir -p gpurun_out
for w in c4 ns64; do
  for ks in 2 3 6 8; do
    echo "== $w new_ks $ks: $(HCV_PRE_NEW_KS=$ks python tools/host_step.py $w 2>&1 | grep -v amdgpu.ids | tail -2 | tr '\n' ' ')"
  done
  echo "== $w off: $(HCV_HOST_PRE_MAC=0 python tools/host_step.py $w 2>&1 | grep -v amdgpu.ids | tail -2 | tr '\n' ' ')"
done 2>&1 | tee gpurun_out/hostpre_ab2.log
