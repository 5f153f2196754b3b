#!/bin/bash
# kernel timeline of the hop-sized steps of one workload: bash tools/micro/trace_w.sh <workload> [ENV=val ...]
# prints the bench line, the per-kernel averages over the last 400 launches and the timeline of the last launches of the timed region
W=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
D=gpurun_out/trace_$W; rm -rf $D
env "$@" timeout 300 rocprofv3 --kernel-trace --output-format csv -d $D -o t -- python bench.py --workload $W --no-cpu-baseline --batched-block 0 --realtime-block 0 --extended-ratio 0 --also "" --steps 64 --warmup 8 --no-self-check $BENCH_ARGS 2>/dev/null < /dev/null | grep '^{' | cut -c1-200
T=$(find $D -name "*kernel_trace.csv" | head -1)
[ -n "$T" ] && python - "$T" <<'PY'
import csv,sys,re,collections
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n=r["Kernel_Name"].replace("void ","").replace("hcv::(anonymous namespace)::","").replace("hcv::","")
    r["n"]=re.sub(r"\(.*","",n)[:46]
rows.sort(key=lambda r:r["s"])
ours=[i for i,r in enumerate(rows) if any(k in r["n"] for k in ("spectral_mac","fused_","mac_meet","rifft","rfft","fwd_publish","reduce_partials","emit"))]
last=ours[-1]
tail=[rows[i] for i in ours[-400:]]
agg=collections.defaultdict(lambda:[0,0.0])
for r in tail:
    a=agg[(r["n"],r["Grid_Size_X"],r["Workgroup_Size_X"])]; a[0]+=1; a[1]+=(r["e"]-r["s"])/1e3
for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1])[:12]: print(f"{k[0]:48s} grid {k[1]:>8s} wg {k[2]:>5s} calls {v[0]:4d} avg {v[1]/v[0]:8.2f} us")
print("--- timeline of the last launches of the timed region")
seg=[rows[i] for i in ours[-int(__import__("os").environ.get("TRACE_ROWS","14")):]]
t0=seg[0]["s"]
for r in seg: print(f"  +{(r['s']-t0)/1e3:8.1f} us dur {(r['e']-r['s'])/1e3:7.1f} end +{(r['e']-t0)/1e3:8.1f} q {r['Queue_Id']:>2s} {r['n']} {r['Grid_Size_X']}")
PY
