#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_race
mkdir -p $O
cd $R
export TMPDIR=/tmp
: > $O/log6.txt
for k in ${KINDS:-swin none matmul aten_ew transpose ln linear mlp96 mlp384 wattn pixdec}; do
  F=/tmp/race6_$k
  rm -f $F.ready $F.done
  timeout 300 python tools/race_probe6.py --role aggressor --kind $k --flag $F --seconds 150 2>&1 | grep -v amdgpu | cut -c1-600 > $O/a6_$k.txt &
  timeout 300 python tools/race_probe6.py --role victim --kind $k --flag $F --iters ${ITERS:-60} 2>&1 | grep -v amdgpu | cut -c1-600 > $O/v6_$k.txt
  wait
  cat $O/v6_$k.txt $O/a6_$k.txt >> $O/log6.txt
done
echo "== one process, two streams" >> $O/log6.txt
for k in swin mlp96 matmul; do
  timeout 300 python tools/race_probe6.py --role both --kind $k --iters 40 2>&1 | grep -v amdgpu | cut -c1-600 >> $O/log6.txt
done
cat $O/log6.txt
