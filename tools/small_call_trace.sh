#!/bin/bash
# Which kernels a paced small-block process() call of a big matrix is made of: tools/small_call_trace.sh <workload> <block> [hops]
# rocprofv3 kernel trace of tools/latency.py, per kernel: launches per call and microseconds per call over the paced phase.
w=${1:-ns64}; B=${2:-32}; hops=${3:-1}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; D=/tmp/sct; rm -rf $D
rocprofv3 --kernel-trace --output-format csv -d $D -- python $R/tools/latency.py $w $B $hops 2>/dev/null | tail -3
python - "$D" $B $hops <<'PY'
import csv, glob, sys, collections
B, hops = int(sys.argv[2]), int(sys.argv[3])
ncalls = hops * 8192 // B
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# the paced phase = everything after the last whole-hop inverse of the priming run
last = max(i for i, r in enumerate(rows) if "rifft_emit" in r[2] or "rifft_split_emit" in r[2])
ph = rows[last + 1:]
agg = collections.defaultdict(lambda: [0, 0.0])
for s, e, k in ph:
    k = k.split("(")[0].replace("void hcv::", "").replace("(anonymous namespace)::", "")[:70]
    agg[k][0] += 1
    agg[k][1] += (e - s) / 1e3
print(f"paced phase: {len(ph)} launches over {ncalls} calls = {len(ph) / ncalls:.1f} per call, {sum(v[1] for v in agg.values()) / ncalls:.1f} us of kernels per call, span {(ph[-1][1] - ph[0][0]) / 1e3 / ncalls:.1f} us per call")
for k, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  {n / ncalls:7.2f} per call  {us / n:8.2f} us each  {us / ncalls:8.2f} us per call  {k}")
PY
