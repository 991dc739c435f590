#!/bin/bash
# round 3, GPU call C: PMC profile of the strips kernel (occupancy, LDS, VALU, waits)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_c
mkdir -p $O
cd $R
export TMPDIR=/tmp
CMD="python tools/kbench.py --only strips"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- $CMD > $O/trace.log 2>&1
python - <<PY > $O/trace_kernel.txt
import csv,glob
for f in glob.glob("$O/trace/**/*kernel_trace.csv", recursive=True):
    rows=[r for r in csv.DictReader(open(f)) if 'strips' in r['Kernel_Name']]
    if rows:
        r=rows[-1]
        print({k:r[k] for k in r if k not in ('Kernel_Name',)})
        d=[int(x['End_Timestamp'])-int(x['Start_Timestamp']) for x in rows]
        print('n',len(d),'avg us',sum(d)/len(d)/1e3,'min',min(d)/1e3)
PY
for i in 1 2 3; do
  case $i in
    1) C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE";;
    2) C="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM";;
    3) C="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES";;
  esac
  rocprofv3 --output-format csv --pmc $C -d $O/pmc$i -o p -- $CMD > $O/pmc$i.log 2>&1
  python tools/pmc_summary.py $O/pmc$i strips > $O/pmc$i.txt 2>&1
done
python tools/kbench.py --only msda 2>/dev/null | grep -E "strips" > $O/kbench_strips.txt
rm -rf $O/pmc1 $O/pmc2 $O/pmc3 $O/trace
echo done
