#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29534
timeout -k 10 400 python -u -m pytorch_distributed_amd.launch -np 1 examples/5_horovod_distributed.py --dataset imagenet --synthetic -a resnet50 -b 400 --precision fp16 --opt-level O2 --epochs 1 --steps-per-epoch 60 -p 20 -j 4 > gpurun_out/example5_r50.log 2>&1 || { tail -n 30 gpurun_out/example5_r50.log | cut -c1-250; exit 2; }
grep -v "^$" gpurun_out/example5_r50.log | tail -n 9 | cut -c1-250
timeout -k 10 400 python -u -m pytest tests/test_examples_gpu.py -m gpu -x -q -k "style5" --timeout 300 --timeout-method thread > gpurun_out/ex5_tests.log 2>&1 || { tail -n 30 gpurun_out/ex5_tests.log | cut -c1-250; exit 3; }
tail -n 2 gpurun_out/ex5_tests.log
