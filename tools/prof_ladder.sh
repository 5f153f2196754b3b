#!/bin/bash
# Kernel trace of the extended far-tail ladder as the headline (bench.py --workload $1 --tail-ratio 8): per-kernel totals of the timed
# window and the timeline of the last few steps (queue, start, duration).  Usage on the box: tools/prof_ladder.sh c5 [ENV=val ...]
w=${1:-c5}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
D=gpurun_out/prof_ladder_$w; rm -rf $D
env "$@" rocprofv3 --kernel-trace --output-format csv -d $D -- python bench.py --workload $w --tail-ratio 8 --no-cpu-baseline --batched-block 0 --realtime-block 0 --also "" --steps 128 --warmup 8 2>/dev/null | grep '^{' | cut -c1-330
T=$(find $D -name "*kernel_trace.csv" | head -1)
python - "$T" <<'PY'
import csv,sys,re,collections
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    r["n"] = re.sub(r"\(.*","",r["Kernel_Name"]).replace("void ","").replace("hcv::","")[:44]
rows.sort(key=lambda r:r["s"])
ems=[r for r in rows if r["n"].startswith("emit_kernel")]
t0=ems[-128]["s"] if len(ems)>=128 else rows[len(rows)//2]["s"]
t1=rows[-1]["e"]
agg=collections.defaultdict(lambda:[0,0.0])
for r in rows:
    if r["s"]<t0: continue
    a=agg[(r["n"],r["Grid_Size_X"],r["Grid_Size_Y"])]; a[0]+=1; a[1]+=(r["e"]-r["s"])/1e3
tot=sum(v[1] for v in agg.values())
print(f"window {(t1-t0)/1e6:.2f} ms ({(t1-t0)/1e3/128:.1f} us per step), kernel time {tot/1e3:.2f} ms")
for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1])[:26]: print(f"{k[0]:46s} grid {k[1]:>8s}x{k[2]:<3s} calls {v[0]:5d} total {v[1]/1e3:8.3f} ms avg {v[1]/v[0]:8.2f} us")
# timeline of three steps in the middle of the window (not a rung boundary if possible)
mid=ems[-40]["e"]; end=ems[-37]["e"]
print("--- timeline of three steps")
for r in rows:
    if r["s"]>=mid and r["s"]<end:
        print(f"  +{(r['s']-mid)/1e3:8.1f} us dur {(r['e']-r['s'])/1e3:7.1f} q{r.get('Queue_Id'):>3} {r['n']} {r['Grid_Size_X']}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}")
PY
rm -rf $D
