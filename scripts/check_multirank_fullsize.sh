#!/bin/bash
# One-off confidence check on a 1-GPU box: bench.py at FULL size with 4 (and 8) ranks sharing GPU 0 (gloo exchanges) must fold to
# the single-rank result.  Usage: bash scripts/check_multirank_fullsize.sh [log_m]
set -u
LM=${1:-22}
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/mr
python bench.py --no-cpu-baseline --steps 1 --warmup 0 --log-m $LM --dump-result gpurun_out/mr/r1.npz > gpurun_out/mr/r1.json || exit 1
for N in 4 8; do
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29700+N)) bench.py --gpus $N --steps 1 --warmup 0 \
      --log-m $LM --no-cpu-baseline --backend gloo --shared-device --dump-result gpurun_out/mr/r$N.npz > gpurun_out/mr/r$N.json 2> gpurun_out/mr/r$N.err || { tail -5 gpurun_out/mr/r$N.err; exit 1; }
  python - <<PY
import numpy as np
a, b = np.load("gpurun_out/mr/r1.npz"), np.load("gpurun_out/mr/r$N.npz")
ok = all(np.array_equal(a[t], b[t]) for t in ("h", "l", "a", "b1", "b2"))
print("world $N at 2^$LM:", "identical to the single-rank result" if ok else "MISMATCH")
PY
done
rm -f gpurun_out/mr/*.npz
