#!/bin/bash
# round 6: counter passes (separate --pmc runs, no trace domains) of the fused MLP on the encoder FFN's shape (tools/ffn_probe.py), final tree
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_pmc_ffn
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
: > $O/ffn_pmc.txt
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" \
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --output-format csv --pmc $C -d $O/pmc_$N -o p -- python $R/tools/ffn_probe.py > $O/pmc_$N.log 2>&1
  python $R/tools/pmc_summary.py $O/pmc_$N mlp_f16x3 >> $O/ffn_pmc.txt 2>&1
  rm -rf $O/pmc_$N
done
cat $O/ffn_pmc.txt
grep -h "us$" $O/pmc_*.log | head -3
