#!/bin/bash
# usage: tools/pmc_mix.sh <tag> <spp> <depth> : VALU instruction mix of k_trace for one rt_render of the bench scene
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; spp=${2:-4}; depth=${3:-1}
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 --output-format csv -d gpurun_out/pmc_${tag}_m -- python3 tools/render_once.py mesh 2 $spp $depth > gpurun_out/pmc_${tag}_m.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT64 SQ_INSTS_SMEM SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS --output-format csv -d gpurun_out/pmc_${tag}_n -- python3 tools/render_once.py mesh 2 $spp $depth > gpurun_out/pmc_${tag}_n.log 2>&1 || exit 1
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(float)
for d in ("m","n"):
    for f in glob.glob("gpurun_out/pmc_${tag}_%s/*/*counter_collection.csv"%d):
        for r in csv.DictReader(open(f)):
            if "k_trace" in r["Kernel_Name"]: acc[r["Counter_Name"]]+=float(r["Counter_Value"])
for k in sorted(acc): print("%-26s %.5g"%(k,acc[k]))
PY
