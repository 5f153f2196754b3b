#!/bin/bash
# kernel timeline of one rank's share of strong-scaled c4 (64 x 8, 8192-sample calls)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
D=gpurun_out/trace_c4s8; rm -rf $D
env "$@" timeout 200 rocprofv3 --kernel-trace --output-format csv -d $D -o t -- python bench.py --workload c4s8 --no-cpu-baseline --batched-block 0 --realtime-block 0 --extended-ratio 0 --also "" --steps 64 --warmup 8 --no-self-check 2>/dev/null < /dev/null | grep '^{' | cut -c1-200
T=$(find $D -name "*kernel_trace.csv" | head -1)
[ -n "$T" ] && python - "$T" <<'PY'
import csv,sys,re,collections
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n=r["Kernel_Name"].replace("void ","").replace("hcv::(anonymous namespace)::","").replace("hcv::","")
    r["n"]=re.sub(r"\(.*","",n)[:46]
rows.sort(key=lambda r:r["s"])
tail=rows[-400:]
agg=collections.defaultdict(lambda:[0,0.0])
for r in tail:
    a=agg[(r["n"],r["Grid_Size_X"],r["Workgroup_Size_X"])]; a[0]+=1; a[1]+=(r["e"]-r["s"])/1e3
for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1])[:12]: print(f"{k[0]:48s} grid {k[1]:>8s} wg {k[2]:>5s} calls {v[0]:4d} avg {v[1]/v[0]:8.2f} us")
print("--- timeline of the last three steps")
em=[i for i,r in enumerate(rows) if r["n"].startswith("emit")]
i0=em[-4]+1 if len(em)>=4 else len(rows)-30
t0=rows[i0]["s"]
for r in rows[i0:]: print(f"  +{(r['s']-t0)/1e3:8.1f} us dur {(r['e']-r['s'])/1e3:7.1f} q {r['Queue_Id']:>2s} {r['n']} {r['Grid_Size_X']}")
PY
