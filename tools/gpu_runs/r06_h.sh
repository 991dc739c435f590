#!/bin/bash
# round 6: generation 6, one order for all waves, commits in front of the point reduction: parity, timing of two tilings x two
# schedules, the phase timeline, kbench
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_h
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
python tools/msda_probe.py --gen 6 --geom cfg5 --T 2 2>/dev/null
python tools/msda_probe.py --gen 5 --geom cfg5 --T 2 2>/dev/null
python tools/msda_probe.py --gen 6 --geom cfg1 --T 2 2>/dev/null
python tools/msda_probe.py --gen 5 --geom cfg1 --T 2 2>/dev/null
python tools/msda_probe.py --gen 6 --T 1 2>/dev/null
python tools/msda_probe.py --gen 5 --T 1 2>/dev/null
UNIVS_HIP_LIB=$R/univs_amd/libunivs_hip_heads_trace.so python tools/msda_trace6.py > $O/trace_lockstep.txt 2>$O/trace_lockstep.err
cat $O/trace_lockstep.txt
timeout 300 python tools/kbench.py --only msda > $O/kbench_msda.json 2> $O/kbench_msda.err
grep -E "heads|strips\"|strips \{|tiled2" $O/kbench_msda.json
