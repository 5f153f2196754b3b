#!/bin/bash
# kernel trace of the extended far-tail ladder as the headline (bench.py --tail-ratio 8)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
D=gpurun_out/prof_ext; rm -rf $D
rocprofv3 --kernel-trace --output-format csv -d $D -- python bench.py --tail-ratio 8 --no-cpu-baseline --batched-block 0 --realtime-block 0 --steps 128 --warmup 8 2>/dev/null | grep '^{' | cut -c1-160
T=$(find $D -name "*kernel_trace.csv" | head -1)
python - "$T" <<'PY'
import csv,sys,re,collections
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# the timed region = the last 128 steps: take the last 40% of the trace by time
t1=int(rows[-1]["End_Timestamp"]); 
# find start of last 128 emit-like launches: use rifft_emit / emit kernels count
agg=collections.defaultdict(lambda:[0,0.0])
ems=[r for r in rows if "emit" in r["Kernel_Name"]]
t0=int(ems[-128]["Start_Timestamp"]) if len(ems)>=128 else int(rows[len(rows)//2]["Start_Timestamp"])
for r in rows:
    if int(r["Start_Timestamp"])<t0: continue
    n=re.sub(r"\(.*","",r["Kernel_Name"]).replace("void ","").replace("hcv::","")[:44]
    a=agg[(n,r["Grid_Size_X"])]; a[0]+=1; a[1]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
tot=sum(v[1] for v in agg.values())
print(f"window {(t1-t0)/1e6:.2f} ms, kernel time {tot/1e3:.2f} ms")
for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1])[:22]: print(f"{k[0]:46s} grid {k[1]:>9s} calls {v[0]:5d} total {v[1]/1e3:8.3f} ms avg {v[1]/v[0]:8.2f} us")
PY
rm -rf $D
