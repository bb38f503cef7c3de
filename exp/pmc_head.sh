#!/bin/bash
# usage: exp/pmc_head.sh LIB TAG
set -e
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/pmch_$2; mkdir -p $O
export VQ_VAE_AMD_LIB=$R/$1
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f -o p -- python3 $R/exp/pmc_head.py > $O/f.log 2>&1
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for p in glob.glob("$O/f/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "search_cyclic" in r["Kernel_Name"] or "search_mfma" in r["Kernel_Name"]: acc[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
for k,v in acc.items(): print("$2", k, len(v), "read GB per launch", sum(v)/len(v)*2*1024/1e9)
PY
find $O -name "*counter_collection.csv" -delete
