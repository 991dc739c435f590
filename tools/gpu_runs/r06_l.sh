#!/bin/bash
# round 6: is test_device_sampler_on_the_gpu's run-to-run difference older than this round?  (the round-5 tree, built here, on this round's box)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R/_r05
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
for i in 1 2; do timeout 600 python -m pytest tests/test_modules_gpu.py -m gpu -q -k "device_sampler" 2>&1 | tail -2; done
