#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export PYTHONPATH=$GRAFT_REPO_ROOT
cd /tmp
for extra in "" "--use-adasum" "--fp16-allreduce"; do
  timeout -k 10 200 python -u -m pytorch_distributed_amd.launch -np 1 $GRAFT_REPO_ROOT/examples/5_2_horovod_mnist.py --epochs 2 $extra > $GRAFT_REPO_ROOT/gpurun_out/mnist.log 2>&1 || { echo "mnist $extra FAILED"; tail -n 25 $GRAFT_REPO_ROOT/gpurun_out/mnist.log | cut -c1-220; exit 1; }
  echo "mnist [$extra]: $(grep -i 'test set' $GRAFT_REPO_ROOT/gpurun_out/mnist.log | tail -n 1 | cut -c1-120)"
done
