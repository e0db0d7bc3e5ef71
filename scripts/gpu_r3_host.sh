#!/bin/bash
# host time per step, plain vs single-rank DDP, with the C ready hooks and with the Python closures; then the
# ResNet-18 CIFAR-shaped bench eager with DDP
set -o pipefail
mkdir -p gpurun_out
for n in 1 0; do
  echo "== PDA_DDP_NATIVE_HOOKS=$n" | tee -a gpurun_out/host_phases_r3.txt
  PDA_DDP_NATIVE_HOOKS=$n timeout -k 10 200 python -u benchmarks/host_phases.py --precision fp16 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/host_phases_r3.txt || exit 1
done
for n in 1 0; do
  echo "== bench resnet18 32x32 b800 fp16 O2 --force-ddp, PDA_DDP_NATIVE_HOOKS=$n" | tee -a gpurun_out/host_phases_r3.txt
  PDA_DDP_NATIVE_HOOKS=$n timeout -k 10 200 python -u bench.py --arch resnet18 --image-size 32 --batch-size 800 --steps 60 --warmup 20 --force-ddp --no-compare-stock 2>/dev/null | grep -o '"value": [0-9.]*, \|"ms_per_step": [0-9.]*' | tr -d '\n' | tee -a gpurun_out/host_phases_r3.txt; echo | tee -a gpurun_out/host_phases_r3.txt
done
