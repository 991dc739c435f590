#!/bin/bash
# round 3, GPU call A: new parity tests (teacher-forced cfg 4 / cfg 3, G2, autocast), steady-state PROMPTED clip profile
# (rocprofv3 kernel trace cut per clip + host cProfile), MSDA kbench baseline.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_a
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 900 python -m pytest tests/test_modules_gpu.py -q -m gpu -s -p no:cacheprovider \
  -k "g2_ or config4 or autocast or config2 or config3" > $O/parity.log 2>&1
echo "pytest rc $?" >> $O/parity.log
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prompted -- python $R/tools/prompted_clip.py --clips 10 > $O/prompted_run.log 2>&1
CSV=$(ls $O/prompted/*/*_kernel_trace.csv | head -1)
python $R/tools/clip_breakdown.py $CSV --last 8 --top 70 > $O/prompted_clip_breakdown.txt 2>&1
python $R/tools/clip_breakdown.py $CSV --skip 3 --last 6 --top 40 > $O/first_clip_breakdown.txt 2>&1
rm -rf $O/prompted
cd $R
timeout 300 python tools/prompted_clip.py --clips 10 > $O/prompted_noprof.log 2>&1
timeout 300 python tools/cprof_prompts.py > $O/prompted_cprof.txt 2>&1
timeout 300 python tools/find_syncs.py > $O/prompted_syncs.txt 2>&1
timeout 300 python tools/kbench.py --only msda > $O/kbench_msda.txt 2>&1
echo done
