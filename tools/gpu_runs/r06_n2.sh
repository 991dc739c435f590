#!/bin/bash
# round 6: the N > 1 paths of bench.py end to end on ONE GPU (all ranks on GPU 0, gloo collectives on device tensors: UNIVS_BENCH_ONE_GPU_DEBUG=1) -- a
# correctness run of the replica line, the frame-sharded clip and the sharded sliding loop (teams at N = 8); the TIMES MEAN NOTHING, the parity figures
# and the byte counters of the sliding loop do
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_n2
mkdir -p $O
cd $R
export TMPDIR=/tmp UNIVS_BENCH_ONE_GPU_DEBUG=1
for N in 2 8; do
  timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29600 + N)) bench.py --gpus $N --steps 2 --warmup 1 > $O/bench_n$N.json 2> $O/bench_n$N.err
  echo "N=$N rc $?"
  python - <<PY
import json
try:
    r = json.loads(open("gpurun_out/r06_n2/bench_n$N.json").read().strip().splitlines()[-1])
    print({k: r.get(k) for k in ("n_gpus", "mask_logit_max_abs_err", "mask_sign_flips")})
    sl = r.get("sliding_clip_loop", {}).get("frame_sharded", {})
    print(" sliding loop, sharded:", {k: sl.get(k) for k in ("ranks_used", "ranks_idle", "window", "bytes_received_rank0")})
except Exception as e:
    print("no json line:", e)
    print(open("gpurun_out/r06_n2/bench_n$N.err").read()[-3000:])
PY
done
