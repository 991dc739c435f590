#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_n2
mkdir -p $O
cd $R
export TMPDIR=/tmp UNIVS_BENCH_ONE_GPU_DEBUG=1
for rep in 1 2 3; do
  for N in 2 3; do
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29700 + N + 10 * rep)) bench.py --gpus $N --steps 3 --warmup 1 --no-sliding-loop > $O/b.json 2> $O/b.err
  python - <<PY
import json
r = json.loads(open("gpurun_out/r06_n2/b.json").read().strip().splitlines()[-1])
print("rep $rep N=$N", {k: r.get(k) for k in ("value", "mask_logit_max_abs_err", "mask_sign_flips")})
PY
  done
done
