#!/bin/bash
# round 3: bench line + rocprofv3 kernel stats of the same command + steady-state clip breakdown + HBM traffic (PMC) of the MSDA kernel
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-v1}
O=$R/gpurun_out/r03_prof_$TAG
mkdir -p $O
cd $R
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_rocprof.json 2> $O/trace.err
find $O/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
CSV=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python $R/tools/clip_breakdown.py $CSV --skip 4 --last 8 --top 60 > $O/clip_breakdown.txt 2>&1
rm -rf $O/trace
cd $R
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  N=$(echo $C | tr ' ' '_')
  timeout 300 rocprofv3 --output-format csv --pmc $C -d $O/pmc_$N -o p -- python tools/kbench.py --only strips > $O/pmc_$N.log 2>&1
  python tools/pmc_summary.py $O/pmc_$N strips > $O/pmc_$N.txt 2>&1
  rm -rf $O/pmc_$N
done
echo done
