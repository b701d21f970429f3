#!/bin/bash
# usage: tools/pmc_traffic.sh <tag> <spp> : HBM-side traffic of k_trace for one rt_render of the bench workload
# (FETCH_SIZE and WRITE_SIZE in separate passes; kilobytes; gfx950 correction per MI355X_MICROARCH.md "HBM": FETCH_SIZE x 2)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; spp=${2:-16}
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_${tag}_f -- python3 tools/render_once.py mesh 2 $spp 5 > gpurun_out/pmc_${tag}_f.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_${tag}_w -- python3 tools/render_once.py mesh 2 $spp 5 > gpurun_out/pmc_${tag}_w.log 2>&1 || exit 1
python3 - <<PY
import csv,glob,collections,json
acc=collections.defaultdict(float); n=collections.defaultdict(int)
for d in ("f","w"):
    for f in glob.glob("gpurun_out/pmc_${tag}_%s/*/*counter_collection.csv"%d):
        for r in csv.DictReader(open(f)):
            if "${KFILTER:-k_trace}" in r["Kernel_Name"]:
                acc[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
L=max(n.get("FETCH_SIZE",1),1)
out={"workload":"pagoda_1080p_brdf","spp_per_step":$spp,"k_trace_launches":L,
     "fetch_size_kb_raw_per_launch":acc["FETCH_SIZE"]/L,"write_size_kb_per_launch":acc["WRITE_SIZE"]/max(n.get("WRITE_SIZE",1),1),
     "hbm_bytes_per_launch":(2*acc["FETCH_SIZE"]/L+acc["WRITE_SIZE"]/max(n.get("WRITE_SIZE",1),1))*1024,
     "note":"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one pass each, k_trace dispatches only, averaged over the launches of one 5-bounce pass; FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B)"}
print(json.dumps(out))
PY
