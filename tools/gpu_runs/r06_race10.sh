#!/bin/bash
# after the single-operation helpers (csrc/common.h): the kernels that changed against their tests, the co-residency probes again, the
# N = 2 bench with both ranks on GPU 0 (parity under sharing), then the whole GPU suite
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_fix
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -x -k "transpose or patch_embed or resample or upsample or tokens or msda or strips or pixel_decoder or swin" > $O/pytest_affected.log 2>&1; echo "affected rc $?"; tail -3 $O/pytest_affected.log
timeout 600 python tools/race_probe7.py --iters 20 2>&1 | grep -v amdgpu | cut -c1-300 > $O/probe7_swin.txt; grep -c "<<<<" $O/probe7_swin.txt; head -3 $O/probe7_swin.txt
for i in 1 2; do timeout 600 python tools/race_probe.py --tag p$i --iters 30 2>&1 | grep -v amdgpu | cut -c1-400 > $O/probe1_$i.txt & done; wait; cat $O/probe1_1.txt $O/probe1_2.txt
export UNIVS_BENCH_ONE_GPU_DEBUG=1
for rep in 1 2; do
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29700 + rep)) bench.py --gpus 2 --steps 3 --warmup 1 --no-sliding-loop > $O/b2.json 2> $O/b2.err
  python - <<PY
import json
r = json.loads(open("gpurun_out/r06_fix/b2.json").read().strip().splitlines()[-1])
print("N=2 on one GPU, rep $rep", {k: r.get(k) for k in ("value", "mask_logit_max_abs_err", "mask_sign_flips")})
PY
done
unset UNIVS_BENCH_ONE_GPU_DEBUG
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "full gpu suite rc $?"; tail -4 $O/pytest_gpu.log
