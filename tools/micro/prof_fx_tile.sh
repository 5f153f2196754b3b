#!/bin/bash
# per-kernel durations of the four-step passes (1 GiB batches) per variant and scratch chunk:  prof_fx_tile.sh "0 1" "64 1024" rows
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/fxtile
mkdir -p $O
ROWS=${3:-fft:f32:20,fft:f32:18,fft:f32:16}
for t in ${1:-0 1}; do
  for c in ${2:-64 1024}; do
    d=$O/prof_t${t}_c${c}; rm -rf $d
    HCV_FX_TILE=$t HCV_FX_CHUNK_MB=$c timeout 120 rocprofv3 --kernel-trace --output-format csv -d $d -o p -- python $R/tests/perf/bench_fft.py --only $ROWS --reps 3 > $O/prof_t${t}_c${c}.log 2>&1 < /dev/null
    echo "== t=$t chunk=$c"
    grep achieved $O/prof_t${t}_c${c}.log | python -c "import sys,json
for l in sys.stdin:
    r=json.loads(l); print('  2^%d %.4f ms %.0f GB/s' % (r['log2n'], r['ms'], r['achieved_GBps']))"
    f=$(find $d -name "*kernel_trace.csv" | head -1)
    [ -n "$f" ] && python - "$f" <<'PY'
import csv,sys,re,collections
agg=collections.defaultdict(lambda:[0,0.0,1e9])
for r in csv.DictReader(open(sys.argv[1])):
    n=re.sub(r"\(.*","",r["Kernel_Name"]).replace("void ","").replace("hcv::(anonymous namespace)::","")[:60]
    if "fx_" not in n: continue
    d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    a=agg[(n,r["Grid_Size_X"],r["Grid_Size_Y"],r["Workgroup_Size_X"])]; a[0]+=1; a[1]+=d; a[2]=min(a[2],d)
for k,v in sorted(agg.items()): print(f"  {k[0]:44s} grid {k[1]:>8s}x{k[2]:<4s} wg {k[3]:>4s} calls {v[0]:4d} avg {v[1]/v[0]:8.2f} us min {v[2]:8.2f}")
PY
  done
done
