#!/bin/bash
# SQ / TCC / TCP counters of one spectral_mac shape, one rocprofv3 --pmc pass per counter set (no other trace domains):
#   tools/micro/pmc_mac2.sh "<mac_bench args>" <outfile>
# Prints, per counter, the average over the timed launches of the kernel with the longest total time.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/$2
: > $out
echo "# mac_bench $1" >> $out
$R/tools/micro/build/mac_bench $1 >> $out
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS" \
           "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  rm -rf /tmp/pm
  timeout 300 rocprofv3 --pmc $set -d /tmp/pm -o run --output-format csv -- $R/tools/micro/build/mac_bench $1 > /tmp/pm.log 2>&1
  python3 - >> $out <<PY
import csv,collections,glob
f=glob.glob('/tmp/pm/**/run_counter_collection.csv', recursive=True)
if not f:
    print("(no counter file for: $set)")
else:
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        if 'spectral_mac' in r['Kernel_Name']: agg[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
    for kn, d in agg.items():
        for k,v in d.items(): print(f"{kn:60s} {k:32s} {sum(v)/len(v):16.1f}  (n={len(v)})")
PY
done
cat $out
