#!/bin/bash
# round 5: PMC counters per kernel over a short bench run (two separate --pmc passes, no trace domains): where the GEMM family waits
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_pmc
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
cd /tmp; export TMPDIR=/tmp
i=0
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 500 rocprofv3 --output-format csv --pmc $C -d $O/p$i -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-config5 --no-frame-sharded --no-config4 --no-sliding-loop > $O/p$i.log 2>&1
  python $R/tools/pmc_summary.py $O/p$i "mlp_f16x3,linear_f16x3,gemm_f16x3_stream,small_linear,small_chain,xattn_partial,msda_fwd_strips,window_attn_img" > $O/pmc_pass$i.txt 2>&1
  rm -rf $O/p$i
done
head -30 $O/pmc_pass1.txt
