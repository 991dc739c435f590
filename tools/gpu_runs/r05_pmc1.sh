#!/bin/bash
# round 5 (final tree): one PMC pass per kernel over a short bench run -- matrix-pipe busy and vector-ALU share of the GEMM kernels after the tiling
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_pmc1
mkdir -p $O
cd $R
cd /tmp; export TMPDIR=/tmp
timeout 110 rocprofv3 --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_VALU -d $O/p1 -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-config5 --no-frame-sharded --no-config4 --no-sliding-loop > $O/p1.log 2>&1
python $R/tools/pmc_summary.py $O/p1 "mlp_f16x3,linear_f16x3,gemm_f16x3_stream,gemm_f16x3_tile,msda_fwd_strips,window_attn_img" > $O/pmc_pass1.txt 2>&1
rm -rf $O/p1
head -60 $O/pmc_pass1.txt
