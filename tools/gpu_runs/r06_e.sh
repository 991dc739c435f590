#!/bin/bash
# round 6: phase timeline of msda_fwd_heads (instrumented build), lockstep and contiguous schedules
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_e
mkdir -p $O
cd $R
export TMPDIR=/tmp
python tools/msda_probe.py --gen 6 2>/dev/null
UNIVS_HIP_LIB=$R/univs_amd/libunivs_hip_heads_trace.so python tools/msda_probe.py --gen 6 2>/dev/null
UNIVS_HIP_LIB=$R/univs_amd/libunivs_hip_heads_trace.so python tools/msda_trace6.py > $O/trace_lockstep.txt 2>$O/trace_lockstep.err
cat $O/trace_lockstep.txt; tail -3 $O/trace_lockstep.err
UNIVS_HIP_LIB=$R/univs_amd/libunivs_hip_heads_trace.so python tools/msda_trace6.py --cfg msda_sched=1 > $O/trace_contig.txt 2>$O/trace_contig.err
cat $O/trace_contig.txt
