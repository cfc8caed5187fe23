#!/bin/bash
# usage (GPU box, repo root): tools/rehearse_tp.sh <ranks> [bench args...]
# First-contact rehearsal of the TP > 1 code path of bench.py on ONE GPU: <ranks> processes share GPU 0
# (MI_BENCH_SAME_GPU=1, gloo for the host-side collectives, the native all-reduce over hipIpc handles to the same
# device).  Not a measurement: it shows that the world-N path (communicator set-up, self-check against the library
# collective, two-batch step, lm_head all-gather) runs end to end before the driver's multi-GPU run meets it.
set -o pipefail
n=$1; shift
export MI_BENCH_SAME_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 420 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus $n --dist-backend gloo --steps 5 --warmup 2 --no-cpu-baseline --prefill-batch 0 --no-plugin-surface "$@"
