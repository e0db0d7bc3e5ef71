#!/bin/bash
# ResNet-18 on CIFAR10-shaped input (the reference's default workload), batch 800, 1 GPU: eager and hipGraph, fp32 and fp16
set -o pipefail
mkdir -p gpurun_out
run() { timeout -k 10 200 python -u bench.py --arch resnet18 --image-size 32 --batch-size 800 --steps 40 --warmup 15 $1 > gpurun_out/c3.log 2>&1 || { tail -n 20 gpurun_out/c3.log; exit 1; }; echo "resnet18 32x32 b800 $1: $(tail -n 1 gpurun_out/c3.log | grep -o '"value": [0-9.]*') img/s $(tail -n 1 gpurun_out/c3.log | grep -o '"ms_per_step": [0-9.]*') $(tail -n 1 gpurun_out/c3.log | grep -o '"host_issue_ms_per_step": [0-9.]*')"; }
run "--precision fp32"
run "--precision fp32 --graph"
run "--precision fp16"
run "--precision fp16 --graph"
run "--precision fp32 --force-ddp"
