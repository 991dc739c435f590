#!/bin/bash
# round 6: generation 6 with two wave groups (X streams while Y does its vector-only work and vice versa), packed piece
# descriptors, branch-free commits: parity, timing beside the one-group build, the phase timeline
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_g
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
UNIVS_HIP_LIB=$R/univs_amd/libunivs_hip_heads_onegroup.so python tools/msda_probe.py --gen 6 2>/dev/null
python tools/msda_probe.py --gen 5 2>/dev/null
done
UNIVS_HIP_LIB=$R/univs_amd/libunivs_hip_heads_trace.so python tools/msda_trace6.py > $O/trace_lockstep.txt 2>$O/trace_lockstep.err
cat $O/trace_lockstep.txt
