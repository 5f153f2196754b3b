#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python tools/host_step.py ${1:-c4}
D=gpurun_out/prof_host; rm -rf $D
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $D -- python tools/host_step.py ${1:-c4} 6 > /dev/null 2>&1
K=$(find $D -name "*kernel_trace.csv" | head -1); M=$(find $D -name "*memory_copy_trace.csv" | head -1)
python - "$K" "$M" <<'PY'
import csv,sys,re
ev=[]
for r in csv.DictReader(open(sys.argv[1])):
    ev.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),re.sub(r"\(.*","",r["Kernel_Name"]).replace("void ","").replace("hcv::","")[:40]))
try:
    for r in csv.DictReader(open(sys.argv[2])):
        ev.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),"COPY "+r.get("Direction","")+" "+r.get("Bytes", r.get("Size","?"))))
except Exception as e: print("no copy trace", e)
ev.sort()
last=ev[-70:]
t0=last[0][0]
for s,e,n in last: print(f"{(s-t0)/1e3:10.1f} us  dur {(e-s)/1e3:8.1f}  {n}")
PY
rm -rf $D
