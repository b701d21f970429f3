#!/bin/bash
# usage: tools/pmc_render.sh <tag> <spp> <depth> : SQ / GRBM counter passes over k_trace for one rt_render of the bench scene
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; spp=${2:-4}; depth=${3:-5}
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_INSTS_BRANCH --output-format csv -d gpurun_out/pmc_${tag}_a -- python3 tools/render_once.py mesh 2 $spp $depth > gpurun_out/pmc_${tag}_a.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_SMEM --output-format csv -d gpurun_out/pmc_${tag}_b -- python3 tools/render_once.py mesh 2 $spp $depth > gpurun_out/pmc_${tag}_b.log 2>&1 || exit 1
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(float); n=collections.defaultdict(int)
for d in ("a","b"):
    for f in glob.glob("gpurun_out/pmc_${tag}_%s/*/*counter_collection.csv"%d):
        for r in csv.DictReader(open(f)):
            if "${KFILTER:-k_trace}" in r["Kernel_Name"]:
                acc[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
w=acc.get("SQ_WAVES",1)
print("k_trace dispatches:", n.get("SQ_WAVES"))
for k in sorted(acc): print("%-24s %.5g  per-wave %.5g"%(k,acc[k],acc[k]/w))
gui=acc.get("GRBM_GUI_ACTIVE",0)/8.0
if gui:
    print("kernel cycles (GRBM_GUI_ACTIVE/8, summed over dispatches): %.4g"%gui)
    print("VALU issue utilisation = SQ_INSTS_VALU*4 / (1024 SIMDs * cycles) = %.3f"%(acc["SQ_INSTS_VALU"]*4/(1024*gui)))
    print("scalar issue per CU = (SALU+SMEM+BRANCH) / (256 CUs * cycles) = %.3f"%((acc["SQ_INSTS_SALU"]+acc["SQ_INSTS_SMEM"]+acc["SQ_INSTS_BRANCH"])/(256*gui)))
PY
