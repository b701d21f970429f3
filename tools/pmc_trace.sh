#!/bin/bash
# usage: tools/pmc_trace.sh <tag> : SQ counter passes over the traversal kernel on 1080p pagoda primary rays
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/pmc_${tag}_a -- python3 tools/prof_trace.py mesh 1920 1080 1 > gpurun_out/pmc_${tag}_a.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_INST_LEVEL_SMEM SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pmc_${tag}_b -- python3 tools/prof_trace.py mesh 1920 1080 1 > gpurun_out/pmc_${tag}_b.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32 --output-format csv -d gpurun_out/pmc_${tag}_c -- python3 tools/prof_trace.py mesh 1920 1080 1 > gpurun_out/pmc_${tag}_c.log 2>&1
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(float)
for d in ("a","b","c"):
    for f in glob.glob("gpurun_out/pmc_${tag}_%s/*/*counter_collection.csv"%d):
        for r in csv.DictReader(open(f)):
            if "k_trace" in r["Kernel_Name"]: acc[r["Counter_Name"]]=acc.get(r["Counter_Name"],0)+float(r["Counter_Value"])
w=acc.get("SQ_WAVES",1)
for k in sorted(acc): print("%-24s %.4g  per-wave %.4g"%(k,acc[k],acc[k]/w))
PY
