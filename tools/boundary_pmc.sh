#!/bin/bash
# HBM traffic of the boundary segment (quant_conv -> filtered quantizer -> post_quant_conv, forward + backward,
# 4 Mi latents): three rocprofv3 --pmc passes of tools/boundary_prof.py, one counter group each, and their summary.
#   tools/boundary_pmc.sh [round-tag]
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-r3}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout -s KILL 240 rocprofv3 --pmc $c --output-format csv -d $OUT/bpmc_$n -o pmc -- python3 $R/tools/boundary_prof.py > $OUT/bpmc_$n.log 2>&1
  echo pmc $n done
done
python3 $R/tools/pmc_summary.py --fetch $OUT/bpmc_FETCH_SIZE --write $OUT/bpmc_WRITE_SIZE --l2 $OUT/bpmc_TCC_HIT_sum --out $OUT/pmc_boundary.json --command "python3 tools/boundary_prof.py"
find $OUT \( -name '*counter_collection.csv' \) -delete
echo boundary pmc done
