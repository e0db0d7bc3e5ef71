#!/bin/bash
# the public training loop (examples + engine.fit) on the benchmark's workload: ResNet-50, fp16 O2, batch 400, synthetic ImageNet
set -o pipefail
mkdir -p gpurun_out
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533
timeout -k 10 400 python -u -m pytorch_distributed_amd.launch --nproc_per_node 1 --master_addr 127.0.0.1 --master_port 29533 examples/2_distributed.py --dataset imagenet --synthetic -a resnet50 -b 400 --precision fp16 --opt-level O2 --epochs 1 --steps-per-epoch 60 -p 20 -j 4 > gpurun_out/example2_r50.log 2>&1 || { tail -n 30 gpurun_out/example2_r50.log | cut -c1-250; exit 1; }
grep -v "^$" gpurun_out/example2_r50.log | tail -n 8 | cut -c1-250
timeout -k 10 400 python -u -m pytorch_distributed_amd.launch -np 1 examples/5_horovod_distributed.py --dataset imagenet --synthetic -a resnet50 -b 400 --precision fp16 --opt-level O2 --epochs 1 --steps-per-epoch 60 -p 20 -j 4 > gpurun_out/example5_r50.log 2>&1 || { tail -n 30 gpurun_out/example5_r50.log | cut -c1-250; exit 2; }
grep -v "^$" gpurun_out/example5_r50.log | tail -n 8 | cut -c1-250
