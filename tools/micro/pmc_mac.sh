#!/bin/bash
# SQ / TCC counters of one spectral_mac shape: tools/micro/pmc_mac.sh "<mac_bench args>"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/pm
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d /tmp/pm -o run --output-format csv -- $R/tools/micro/build/mac_bench $1 > /dev/null 2>&1
  python3 - <<PY
import csv,collections,glob
f=glob.glob('/tmp/pm/**/run_counter_collection.csv', recursive=True)
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if 'spectral_mac' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items(): print(k, sum(v)/len(v))
PY
done
