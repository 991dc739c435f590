#!/bin/bash
# round 6: generation 6 with the row requests spread over the gather stream, the LDS commits in front of the point reduction, the
# rebalanced remainder round: parity, timing (two tilings, two schedules), the phase timeline
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_f
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "heads" > $O/pytest_heads.log 2>&1
tail -3 $O/pytest_heads.log
for i in 1 2; do
python tools/msda_probe.py --gen 6 2>/dev/null
python tools/msda_probe.py --gen 6 --cfg msda_sched=1 2>/dev/null
python tools/msda_probe.py --gen 6 --cfg msda_strip_w=16,msda_strip_h=6 2>/dev/null
python tools/msda_probe.py --gen 6 --cfg msda_sched=1,msda_strip_w=16,msda_strip_h=6 2>/dev/null
python tools/msda_probe.py --gen 5 2>/dev/null
done
UNIVS_HIP_LIB=$R/univs_amd/libunivs_hip_heads_trace.so python tools/msda_trace6.py > $O/trace_lockstep.txt 2>$O/trace_lockstep.err
cat $O/trace_lockstep.txt
UNIVS_HIP_LIB=$R/univs_amd/libunivs_hip_heads_trace.so python tools/msda_trace6.py --cfg msda_strip_w=16,msda_strip_h=6 > $O/trace_w16h6.txt 2>$O/trace_w16h6.err
cat $O/trace_w16h6.txt
