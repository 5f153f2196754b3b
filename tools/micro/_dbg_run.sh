run() { W=$1; shift; echo "=== $W $*"; bash tools/micro/trace_w.sh $W "$@" 2>&1 | grep -v "rfft_ir\|true, false\|rfft_frames_direct\|rifft_split\|reduce_part" | head -16 | cut -c1-160; }
run c4s8 HCV_NXM_DBG=2
run c4s8 HCV_NXM_DBG=0
