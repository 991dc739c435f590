#!/bin/bash
# round 6: generation 6 -- where the time and the bytes go.  One variant per process; separate --pmc passes.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_c
mkdir -p $O
cd $R
export TMPDIR=/tmp
run() {  # tag, probe args
  TAG=$1; shift
  python tools/msda_probe.py "$@" > $O/${TAG}_time.json 2>/dev/null
  cat $O/${TAG}_time.json
  for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_SMEM"; do
    N=$(echo $C | tr ' ' '_' | cut -c1-40)
    timeout 200 rocprofv3 --output-format csv --pmc $C -d $O/pmc_${TAG}_$N -o p -- python tools/msda_probe.py "$@" > $O/pmc_${TAG}_$N.log 2>&1
    python tools/pmc_summary.py $O/pmc_${TAG}_$N msda_fwd >> $O/${TAG}_pmc.txt 2>&1
    rm -rf $O/pmc_${TAG}_$N
  done
  cat $O/${TAG}_pmc.txt
}
run heads_lockstep --gen 6
run heads_contig --gen 6 --cfg msda_sched=1
run heads_w16h6_contig --gen 6 --cfg msda_sched=1,msda_strip_w=16,msda_strip_h=6
run strips --gen 5
