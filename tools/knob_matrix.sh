#!/bin/bash
# The scheduling knobs change the schedule, never the result: the parity, steady-state and restart suites under each of them.
# Usage on the GPU box: tools/knob_matrix.sh   (one line per knob setting);  KNOBS="A=1|B=0 C=1" tools/knob_matrix.sh runs just those.
# (HCV_FFT_SPLIT=0 on its own is not in the list: the fused block keeps its residue-split transforms, so serial blocks and pipelined
#  blocks would then factorise their transforms differently and the bit-for-bit comparison of the two — not the parity — fails.)
LIST=(HCV_DEFER=0 HCV_TAIL_HEAD=0 HCV_TAIL_GATE=1 HCV_SERIAL=0 HCV_SERIAL=1 HCV_PIPE2=1 HCV_PIPE2=0 HCV_ZERO_COPY=0
      HCV_COOP=0 HCV_COOP_SPIN=0 HCV_FFT_SPLIT=1 "HCV_COOP=0 HCV_FFT_SPLIT=0" HCV_PIVOT_LANES=1 HCV_QUEUE_PROBE=0 HCV_CTL_RESERVE_MB=0 HCV_ORDER_CHECK=1
      HCV_MAC_MFMA=0 HCV_HOST_PRE_MAC=0 HCV_ROCTX=1)
if [ -n "$KNOBS" ]; then IFS='|' read -ra LIST <<< "$KNOBS"; fi
for kv in "${LIST[@]}"; do
  echo -n "$kv: "
  # (round 6: the offline suite too — its statistics expect the matrix-core kernel on whole-hop blocks, so not where a knob takes either away)
  OFF=tests/test_offline_mfma_gpu.py
  case "$kv" in HCV_MAC_MFMA=0|HCV_TAIL_HEAD=0) OFF="";; esac
  env $kv python -m pytest tests/test_gpu_parity.py tests/test_pair_restart_gpu.py tests/test_steady_state_gpu.py tests/test_restart_golden.py tests/test_small_engine_pipeline_gpu.py tests/test_configs_dense_gpu.py $OFF -q -m gpu 2>&1 | grep -E "passed|failed|^FAILED" | tail -4
done
