#!/bin/bash
# full GPU suite + smoke + bench (what the driver runs at round end)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-full}
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -q -m gpu -s -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
echo "smoke rc $?" >> $O/smoke.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
echo done
