#!/bin/bash
# round 6: baseline of the tree as round 5 left it (kbench strips, the bench line) on this round's box
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_a
mkdir -p $O
cd $R
export TMPDIR=/tmp
python tools/kbench.py --only strips > $O/kbench_strips.json 2> $O/kbench_strips.err
cat $O/kbench_strips.json
python bench.py > $O/bench.json 2> $O/bench.err
tail -c 3000 $O/bench.json
