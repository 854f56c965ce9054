#!/bin/bash
# repeat the two-rank worker (tests/dp_gpu_worker.py, "full") and print the DP-vs-single distances.  usage: tools/dp_flake.sh N [ENV=V ...]
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd $GRAFT_REPO_ROOT
N=$1; shift
for i in $(seq 1 $N); do
  env "$@" MASTER_ADDR=127.0.0.1 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port $((29600 + i)) tests/dp_gpu_worker.py $GRAFT_REPO_ROOT full 2>&1 | grep -a "identical shards\|different shards\|mean-gradient\|diverged" | tr '\n' ';'
  echo
done
