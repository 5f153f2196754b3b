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
# timeline of individual calls: a call's launches end with its emit_kernel
segs, cur = [], []
for r in ph:
    cur.append(r)
    if "emit_kernel" in r[2]:
        segs.append(cur); cur = []
def short(k): return k.split("(")[0].replace("void hcv::", "").replace("(anonymous namespace)::", "")[:60]
order = sorted(range(len(segs)), key=lambda i: -(segs[i][-1][1] - segs[i][0][0]))
for idx in order[3:5] + order[len(order) // 2: len(order) // 2 + 1]:
    sg = segs[idx]
    t0 = sg[0][0]
    print(f"call {idx}: {len(sg)} launches, {(sg[-1][1] - t0) / 1e3:.1f} us from its first kernel's start to the end of its emit")
    for st_, e_, k in sg:
        print(f"    +{(st_ - t0) / 1e3:8.1f} us  {(e_ - st_) / 1e3:7.1f} us  {short(k)}")
PY
