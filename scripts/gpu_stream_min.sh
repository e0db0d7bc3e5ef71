#!/bin/bash
# size threshold of the non-temporal stores (PDA_STREAM_MIN_MB): ResNet-50 benchmark and two small-activation models
set -o pipefail
mkdir -p gpurun_out
for rep in 1 2; do
for mb in 0 64 200 100000; do
  PDA_STREAM_MIN_MB=$mb timeout -k 10 300 python -u bench.py --steps 20 --warmup 10 > gpurun_out/sm.log 2>&1 || { tail -n 20 gpurun_out/sm.log; exit 1; }
  echo "resnet50 b400, stream >= $mb MB: $(tail -n 1 gpurun_out/sm.log | grep -o '"ms_per_step": [0-9.]*')"
done
done
for mb in 0 64 100000; do
  PDA_STREAM_MIN_MB=$mb timeout -k 10 300 python -u benchmarks/zoo_bench.py --batch 128 --archs alexnet,resnet18,inception_v3 2>&1 | grep "img/s" | sed "s/^/stream >= $mb MB: /"
done
