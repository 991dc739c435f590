#!/bin/bash
# round 4, GPU call I: small-M Linear heuristics, cheaper wrapper calls (raw stream, no device context, integer pointers),
# T=10 / Q'=200 head golden; full GPU suite, bench, host profile
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_i
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 1800 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/gputests.log 2>&1
echo "pytest rc $?" >> $O/gputests.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config5 --no-frame-sharded > $O/bench.json 2> $O/bench.err
timeout 300 python tools/cprof_step.py > $O/cprof_step.txt 2>&1
echo done
