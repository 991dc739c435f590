#!/bin/bash
# round 6: the prompted clip: timing + kernel trace cut per clip (launches, busy, idle), with the fused attention gate at S >= $1
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_x
mkdir -p $O
cd $R
export TMPDIR=/tmp
python tools/prompted_clip.py --clips 10 2>&1 | grep -v amdgpu
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/tracep -o t -- python $R/tools/prompted_clip.py --clips 8 > $O/prompted.log 2> $O/tracep.err
CSV=$(find $O/tracep -name "*kernel_trace.csv" | head -1)
python $R/tools/clip_breakdown.py $CSV --last 6 --top 90 > $O/prompted_clip_breakdown.txt 2>&1
python $R/tools/clip_breakdown.py $CSV --last 1 --timeline > $O/prompted_clip_timeline.txt 2>&1
rm -rf $O/tracep
head -60 $O/prompted_clip_breakdown.txt | cut -c1-150
