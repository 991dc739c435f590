#!/bin/bash
# round 4: the whole GPU suite on the final tree (what the driver runs at round end), then smoke()
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_full
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/gpu_tests.log 2>&1
echo "pytest rc $?" >> $O/gpu_tests.log
tail -15 $O/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
