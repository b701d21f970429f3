#!/bin/bash
# one gpurun call: per-wave instruction counts of k_trace with and without its walk (tools/setup_cost.py) -> gpurun_out/<tag>_setup_cost.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; o=gpurun_out/$tag; out=${o}_setup_cost.txt; : > $out
for mode in full miss; do
  timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS --output-format csv -d ${o}_sc_$mode -- python3 tools/setup_cost.py $mode > ${o}_sc_$mode.log 2>&1 || { tail -5 ${o}_sc_$mode.log; exit 1; }
  python3 - ${o}_sc_$mode $mode >> $out <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "k_trace" not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES": calls[k] += 1
for k, c in acc.items():
    w = c["SQ_WAVES"]
    print("%-5s %-44s launches %d waves/launch %.0f | per wave: VALU %.0f SALU %.0f SMEM %.0f LDS %.0f | wave cycles/wave %.0f  VALU-active cycles/wave %.0f" % (sys.argv[2], k[:44], calls[k], w / calls[k], c["SQ_INSTS_VALU"] / w, c["SQ_INSTS_SALU"] / w, c["SQ_INSTS_SMEM"] / w, c["SQ_INSTS_LDS"] / w, c["SQ_WAVE_CYCLES"] / w, c["SQ_ACTIVE_INST_VALU"] / w))
PY
done
cat $out
