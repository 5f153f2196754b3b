#!/bin/bash
# Per-kernel durations of chosen rows of the transform surface: tools/fft_trace.sh "fft:f32:16,fft:f32:20" [tag]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; ROWS=${1:-fft:f32:16}; TAG=${2:-fft}
D=/tmp/fft_trace_$TAG; rm -rf $D
rocprofv3 --kernel-trace --output-format csv -d $D -- python $R/tests/perf/bench_fft.py --reps 3 --only "$ROWS" 2>/dev/null | cut -c1-170
python - "$D" > $R/gpurun_out/fft_trace_$TAG.txt <<'PY'
import csv, glob, sys, collections
rows = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows[r["Kernel_Name"][:110]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print(f"{len(v):6d} launches  avg {sum(v)/len(v):9.2f} us  median {v[len(v)//2]:9.2f} us  total {sum(v)/1e3:9.3f} ms  {k}")
PY
cat $R/gpurun_out/fft_trace_$TAG.txt
