#!/bin/bash
# round 6: generation 6 after the one-round-trip cold start and the reduction in front of barrier A: parity, timing, and the
# instrumented builds (python -m univs_amd.build --ablate heads_*): what each phase of an item costs
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_d
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "heads" > $O/pytest_heads.log 2>&1
tail -3 $O/pytest_heads.log
for i in 1 2; do
python tools/msda_probe.py --gen 6 2>/dev/null
python tools/msda_probe.py --gen 6 --cfg msda_sched=1 2>/dev/null
python tools/msda_probe.py --gen 6 --cfg msda_sched=1,msda_strip_w=16,msda_strip_h=6 2>/dev/null
python tools/msda_probe.py --gen 5 2>/dev/null
done
for A in heads_nostream heads_norows heads_noreduce heads_norecords heads_onlyrows heads_skeleton; do
  echo $A; UNIVS_HIP_LIB=$R/univs_amd/libunivs_hip_$A.so python tools/msda_probe.py --gen 6 2>/dev/null
done
