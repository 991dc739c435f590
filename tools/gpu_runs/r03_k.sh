#!/bin/bash
# PMC profile of the W-stationary Linears (six-product bf16 and three-product fp16 kernels) in tools/kbench.py --only linear
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_k
mkdir -p $O
cd $R
export TMPDIR=/tmp
CMD="python tools/kbench.py --only linear"
for i in 1 2 3 4; do
  case $i in
    1) C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE";;
    2) C="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM";;
    3) C="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE";;
    4) C="TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCP_TCC_READ_REQ_sum";;
  esac
  timeout 600 rocprofv3 --output-format csv --pmc $C -d $O/pmc$i -o p -- $CMD > $O/pmc$i.log 2>&1
  python tools/pmc_summary.py $O/pmc$i linear_ > $O/pmc$i.txt 2>&1
done
rm -rf $O/pmc1 $O/pmc2 $O/pmc3 $O/pmc4
echo done
