#!/bin/bash
# usage: tools/pmc_scene.sh <tag> <scene> <integrator> <spp> <depth> : instruction counters of every k_trace / k_shade dispatch of one rt_render (render_once.py, 1080p, one lane)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; R="python3 tools/render_once.py $2 $3 $4 $5"
timeout -s KILL 90 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/pmc_${tag}_a -- $R > gpurun_out/pmc_${tag}_a.log 2>&1 || { tail -3 gpurun_out/pmc_${tag}_a.log; exit 1; }
timeout -s KILL 90 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_${tag}_b -- $R > gpurun_out/pmc_${tag}_b.log 2>&1 || { tail -3 gpurun_out/pmc_${tag}_b.log; exit 1; }
python3 - <<PY
import csv,glob,collections
by=collections.OrderedDict()
for d in ("a","b"):
    for f in glob.glob("gpurun_out/pmc_${tag}_%s/**/*counter_collection.csv"%d, recursive=True):
        for r in csv.DictReader(open(f)):
            n=r["Kernel_Name"].replace("void ","")
            if n.startswith("k_trace") or n.startswith("k_shade"):
                by.setdefault((int(r["Dispatch_Id"]), n.split("(")[0][:34]),{})[r["Counter_Name"]]=float(r["Counter_Value"])
for (i,n),c in sorted(by.items()):
    w=max(c.get("SQ_WAVES",1),1)
    print("%4d %-34s waves %8d | per wave: VALU %7.0f SALU %7.0f SMEM %6.0f LDS %6.0f BR %6.0f VMEM rd %6.0f wr %6.0f flat %5.0f | wave cycles/wave %9.0f wait %.2f"%(i,n,w,c.get("SQ_INSTS_VALU",0)/w,c.get("SQ_INSTS_SALU",0)/w,c.get("SQ_INSTS_SMEM",0)/w,c.get("SQ_INSTS_LDS",0)/w,c.get("SQ_INSTS_BRANCH",0)/w,c.get("SQ_INSTS_VMEM_RD",0)/w,c.get("SQ_INSTS_VMEM_WR",0)/w,c.get("SQ_INSTS_FLAT",0)/w,c.get("SQ_WAVE_CYCLES",0)/w,c.get("SQ_WAIT_INST_ANY",0)/max(c.get("SQ_ACTIVE_INST_ANY",1)+c.get("SQ_WAIT_INST_ANY",0),1)))
PY
rm -rf gpurun_out/pmc_${tag}_a gpurun_out/pmc_${tag}_b
