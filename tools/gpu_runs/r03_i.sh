#!/bin/bash
# prompted steady-state clip after the sampler prefetch: kernel trace cut per clip, host cProfile, sync finder
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_i
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prompted -- python $R/tools/prompted_clip.py --clips 10 > $O/prompted_run.log 2>&1
CSV=$(ls $O/prompted/*/*_kernel_trace.csv | head -1)
python $R/tools/clip_breakdown.py $CSV --last 8 --top 200 > $O/prompted_clip_breakdown.txt 2>&1
rm -rf $O/prompted
cd $R
timeout 300 python tools/prompted_clip.py --clips 10 > $O/prompted_noprof.log 2>&1
timeout 300 python tools/cprof_prompts.py > $O/prompted_cprof.txt 2>&1
timeout 300 python tools/find_syncs.py > $O/prompted_syncs.txt 2>&1
echo done
