#!/usr/bin/env bash
# Same CLI as the reference's tools/dist_train.sh:  dist_train.sh <config> <gpus> [train.py args]
# One process per MI355X, RCCL over xGMI (backend "nccl" on ROCm).
CONFIG=$1
GPUS=$2
PORT=${PORT:-29500}
export HSA_ENABLE_IPC_MODE_LEGACY=0
PYTHONPATH="$(dirname $0)/..":$PYTHONPATH \
python -m torch.distributed.run --nnodes=1 --nproc-per-node=$GPUS --master-addr 127.0.0.1 --master-port $PORT \
    $(dirname "$0")/train.py $CONFIG --launcher pytorch ${@:3}
