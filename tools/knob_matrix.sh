#!/bin/bash
# The scheduling knobs change the schedule, never the result: the parity, steady-state and restart suites under each of them.
# Usage on the GPU box: tools/knob_matrix.sh   (one line per knob setting)
# (HCV_FFT_SPLIT=0 on its own is not in the list: the fused block keeps its residue-split transforms, so serial blocks and pipelined
#  blocks would then factorise their transforms differently and the bit-for-bit comparison of the two — not the parity — fails.)
for kv in HCV_DEFER=0 HCV_TAIL_HEAD=0 HCV_HEAD_FFT=0 HCV_PIPELINE=0 HCV_TAIL_GATE=1 HCV_BG_SLICES=3 HCV_ONE_STREAM=1 HCV_SERIAL=0 HCV_SERIAL=1 \
          HCV_DIRECT_IN=0 HCV_DIRECT_OUT=0 HCV_FOLD_REDUCE=0 HCV_PIPE2=1 HCV_PIPE2=0 "HCV_PIPE2=1 HCV_PIPE3=1" "HCV_PIPE2=1 HCV_PIPE_SPARSE=0" "HCV_PIPE2=1 HCV_PIPE_DEPTH=1" HCV_MAC_PREFETCH=0 HCV_ZERO_COPY=0 HCV_SERIAL_KSPLIT=8 \
          HCV_COOP=0 HCV_COOP_HOPS=0 HCV_FFT_SPLIT=1 "HCV_COOP=0 HCV_FFT_SPLIT=0" HCV_MAC_INWG=0 HCV_FUSE_REDUCE=1 HCV_XCD_PIN=0 HCV_REDUCE_FAST=0 HCV_FIR_SMALL=0 HCV_BOUNDARY_KSPLIT=1 HCV_BG_LEAD=0 HCV_BG_LEAD=448 HCV_SERIAL_SMALL=512; do
  echo -n "$kv: "
  env $kv python -m pytest tests/test_gpu_parity.py tests/test_pair_restart_gpu.py tests/test_steady_state_gpu.py tests/test_restart_golden.py tests/test_small_engine_pipeline_gpu.py tests/test_configs_dense_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -1
done
