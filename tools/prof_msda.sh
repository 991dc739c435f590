#!/bin/bash
# usage: tools/prof_msda.sh <outdir>   (run on the GPU box; PMC passes are separate from the trace pass)
set -u
OUT=${1:-gpurun_out/prof_msda}
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python tools/kbench.py --only msda"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
for i in 1 2 3 4; do
  case $i in
    1) C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE";;
    2) C="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM";;
    3) C="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VMEM_WR SQ_INSTS_SMEM";;
    4) C="TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum";;
  esac
  rocprofv3 --output-format csv --pmc $C -d $OUT/pmc$i -o p -- $CMD > $OUT/pmc$i.log 2>&1
  python tools/pmc_summary.py $OUT/pmc$i msda > $OUT/pmc$i.txt 2>&1
done
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/pmc5 -o p -- $CMD > $OUT/pmc5.log 2>&1; python tools/pmc_summary.py $OUT/pmc5 msda > $OUT/pmc5.txt 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/pmc6 -o p -- $CMD > $OUT/pmc6.log 2>&1; python tools/pmc_summary.py $OUT/pmc6 msda > $OUT/pmc6.txt 2>&1
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs cat | head -12 > $OUT/kernel_stats.txt
cat $OUT/kernel_stats.txt $OUT/pmc*.txt
# keep only summaries (the raw csvs are large)
rm -rf $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 $OUT/pmc4 $OUT/pmc5 $OUT/pmc6 $OUT/trace
