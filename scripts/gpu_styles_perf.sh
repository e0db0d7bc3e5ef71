#!/bin/bash
# per-launch-style step time on one GPU (ResNet-50, synthetic ImageNet shape, fp16, batch 256)
set -o pipefail
mkdir -p gpurun_out
C="--arch resnet50 --dataset imagenet --synthetic -b 256 --epochs 1 --steps-per-epoch 40 --precision fp16 -p 10 --workers 0"
L="python -m pytorch_distributed_amd.launch"
export PYTHONPATH=$GRAFT_REPO_ROOT; cd examples
timeout -k 10 280 $L --nproc_per_node=1 --master_port 29611 --use_argv_local_rank 2_distributed.py $C > ../gpurun_out/style2.log 2>&1 || { tail -n 20 ../gpurun_out/style2.log; exit 2; }
timeout -k 10 280 $L --nproc_per_node=1 --master_port 29612 --use_argv_local_rank 4_amp_distributed.py $C > ../gpurun_out/style4.log 2>&1 || { tail -n 20 ../gpurun_out/style4.log; exit 2; }
timeout -k 10 280 $L -np 1 -H localhost:1 5_horovod_distributed.py $C > ../gpurun_out/style5.log 2>&1 || { tail -n 20 ../gpurun_out/style5.log; exit 2; }
cd ..
for s in 2 4 5; do echo "style $s: $(grep 'Epoch: \[0\]\[30/' gpurun_out/style$s.log | tail -n 1 | cut -c1-160)"; done
