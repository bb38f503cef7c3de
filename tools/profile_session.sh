#!/bin/bash
# One profiling session on the GPU box: bench line, rocprofv3 kernel stats, three PMC passes, side timings.
# Every GPU step runs under its own time limit; the script stops at the first failure.
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r2
mkdir -p $OUT
cd $R
timeout -k 10 300 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -o bench -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/prof_bench.log 2>&1
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout -s KILL 240 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$n -o pmc -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-filtered > $OUT/pmc_$n.log 2>&1
done
cd $R
timeout -k 10 300 python3 tools/other_configs.py > $OUT/other_configs.log 2>&1
timeout -k 10 200 python3 tools/conv_bench.py > $OUT/conv_bench.log 2>&1
echo session done
