#!/bin/bash
# One profiling session on the GPU box: bench line, rocprofv3 kernel stats, three PMC passes, side timings.
# Every GPU step runs under its own time limit; the script stops at the first failure.
#   tools/profile_session.sh [round-tag]        (outputs under gpurun_out/<round-tag>/, default r3)
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-r3}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout -k 10 300 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
echo bench done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -o bench -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/prof_bench.log 2>&1
python3 $R/tools/trace_summary.py $OUT/prof_bench $OUT/bench_kernels_by_grid.csv
echo trace done
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout -s KILL 240 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$n -o pmc -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-filtered > $OUT/pmc_$n.log 2>&1
  echo pmc $n done
done
python3 $R/tools/pmc_summary.py --fetch $OUT/pmc_FETCH_SIZE --write $OUT/pmc_WRITE_SIZE --l2 $OUT/pmc_TCC_HIT_sum --out $OUT/pmc_traffic.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_dp -o dp -- python3 $R/tools/dp_step_trace.py 20 > $OUT/dp_step_trace.log 2>&1
python3 $R/tools/trace_summary.py $OUT/prof_dp $OUT/dp_kernels_by_grid.csv --prefix ""
echo dp trace done
cd $R
timeout -k 10 300 python3 tools/other_configs.py > $OUT/other_configs.log 2>&1
echo other configs done
timeout -k 10 300 python3 tools/shape_sweep.py > $OUT/shape_sweep.log 2>&1
echo shape sweep done
timeout -k 10 200 python3 tools/torch_formulation_gpu.py > $OUT/torch_formulation_gpu.log 2>&1
timeout -k 10 200 python3 tools/conv_bench.py > $OUT/conv_bench.log 2>&1
# the raw traces are large; the summaries above are what is kept
find $OUT \( -name '*kernel_trace.csv' -o -name '*counter_collection.csv' \) -delete
echo session done
