#!/bin/bash
# round 4: bench line + rocprofv3 kernel stats of the same command + steady-state clip breakdown + HBM traffic (PMC) of the MSDA kernel
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-v1}
O=$R/gpurun_out/r04_prof_$TAG
mkdir -p $O
cd $R
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_rocprof.json 2> $O/trace.err
find $O/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
CSV=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python $R/tools/clip_breakdown.py $CSV --skip 4 --last 8 --top 60 > $O/clip_breakdown.txt 2>&1
python $R/tools/clip_breakdown.py $CSV --skip 6 --last 1 --timeline > $O/clip_timeline.txt 2>&1
rm -rf $O/trace
cd $R
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  N=$(echo $C | tr ' ' '_')
  timeout 300 rocprofv3 --output-format csv --pmc $C -d $O/pmc_$N -o p -- python tools/kbench.py --only strips > $O/pmc_$N.log 2>&1
  python tools/pmc_summary.py $O/pmc_$N strips > $O/pmc_$N.txt 2>&1
  rm -rf $O/pmc_$N
done
# the fused MLP and the cross-attention kernels: matrix-pipe busy cycles, LDS activity / conflicts, where the waves wait
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  N=$(echo $C | cut -d' ' -f1)
  timeout 300 rocprofv3 --output-format csv --pmc $C -d $O/pmcmlp_$N -o p -- python tools/kbench.py --only mlp > $O/pmcmlp_$N.log 2>&1
  python tools/pmc_summary.py $O/pmcmlp_$N mlp_f16x3 > $O/pmcmlp_$N.txt 2>&1
  rm -rf $O/pmcmlp_$N
done
echo done
