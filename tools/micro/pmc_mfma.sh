#!/bin/bash
# Round 6: what bounds the offline multiply-accumulate on the matrix cores (hcv_mac_mfma.hip), from counters of the launch alone
# (tools/micro/build/mac_mfma_check 16 16 704 64: c5's shape, random operands).  One rocprofv3 --pmc pass per set, kernel trace only.
#   GRBM_GUI_ACTIVE / wall time = the clock the launch actually ran at (MI355X_MICROARCH.md, "DVFS give-back": the chip clocks to its power budget)
#   SQ_VALU_MFMA_BUSY_CYCLES against SQ_BUSY_CU_CYCLES / SQ_WAVE_CYCLES: how busy the matrix pipe was while the launch ran
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/pmc_mfma; rm -rf $out; mkdir -p $out
B="$R/tools/micro/build/mac_mfma_check ${SHAPE:-16 16 704 64} 4 8192 0"
timeout 300 $B > $out/plain.log 2>&1
for set in "GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $set --kernel-trace -d $out/$tag -o run --output-format csv -- $B > $out/$tag.log 2>&1
done
python3 - $out <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/run_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "")).replace("void ", "").replace("hcv::", "")
        if "spectral_mac" in n:
            agg[n[:44]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob(out + "/GRBM_GUI_ACTIVE/run_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "")).replace("void ", "").replace("hcv::", "")
        if "spectral_mac" in n:
            dur[n[:44]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(open(out + "/plain.log").read())
for k, c in agg.items():
    print(k, "launches under counters:", {n: len(v) for n, v in c.items()})
    d = sum(dur[k]) / max(1, len(dur[k]))
    for n, v in sorted(c.items()):
        print(f"   {n:32s} avg {sum(v)/len(v):16.1f}")
    if "GRBM_GUI_ACTIVE" in c and d:
        print(f"   launch {d:.1f} us under the GRBM pass -> effective clock {sum(c['GRBM_GUI_ACTIVE'])/len(c['GRBM_GUI_ACTIVE'])/d/1e3:.3f} GHz")
PY
