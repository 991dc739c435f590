#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_race
mkdir -p $O
cd $R
export TMPDIR=/tmp
echo "== probe4 alone" > $O/log.txt
python tools/race_probe4.py --tag alone --iters 20 2>&1 | grep -v amdgpu | cut -c1-400 >> $O/log.txt
echo "== probe4 two processes" >> $O/log.txt
for i in 1 2; do python tools/race_probe4.py --tag p$i --iters 30 2>&1 | grep -v amdgpu | cut -c1-400 > $O/p4_$i.txt & done; wait
cat $O/p4_1.txt $O/p4_2.txt >> $O/log.txt
echo "== probe4 two processes, --sync" >> $O/log.txt
for i in 1 2; do python tools/race_probe4.py --tag p$i --iters 30 --sync 2>&1 | grep -v amdgpu | cut -c1-400 > $O/p4s_$i.txt & done; wait
cat $O/p4s_1.txt $O/p4s_2.txt >> $O/log.txt
for f in swin matmul none; do
echo "== probe5 two processes, filler $f" >> $O/log.txt
for i in 1 2; do python tools/race_probe5.py --tag p$i --iters 30 --filler $f 2>&1 | grep -v amdgpu | cut -c1-400 > $O/p5_$i.txt & done; wait
cat $O/p5_1.txt $O/p5_2.txt >> $O/log.txt
done
cat $O/log.txt
