#!/bin/bash
# usage: tools/prof_split.sh <outdir> [mask|linear]   (run on the GPU box)
# Kernel trace + PMC passes (each set in its own run: never combined with trace domains) for the split-bf16 kernels
# (csrc/mask_decode.hip: skinny_gemm_bf16x6_n32, csrc/linear_split.hip: linear_bf16x6) at the config-2 shapes of
# tools/kbench.py.  What to read: SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x kernel cycles) = matrix-pipe
# utilisation (r01: ~45-50 % by ablation), SQ_WAIT_INST_ANY vs SQ_WAVE_CYCLES (quad-cycles) = where the waves sit,
# FETCH_SIZE / WRITE_SIZE = HBM bytes against the algorithmic 419.7 MB (mask decode) per launch.
set -u
OUT=${1:-gpurun_out/prof_split}
WHAT=${2:-mask}
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python tools/kbench.py --only $WHAT"
FILT="bf16x6,f16x3,skinny_gemm"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" \
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i + 1))
  rocprofv3 --output-format csv --pmc $C -d $OUT/pmc$i -o p -- $CMD > $OUT/pmc$i.log 2>&1
  python tools/pmc_summary.py $OUT/pmc$i $FILT > $OUT/pmc$i.txt 2>&1
done
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs cat | head -14 > $OUT/kernel_stats.txt
cat $OUT/kernel_stats.txt $OUT/pmc*.txt
rm -rf $OUT/pmc[0-9] $OUT/trace      # keep only the summaries (the raw csvs are large)
