#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for k in 0 128 256 576 1152; do
PDA_IGEMM_SHALLOW_K=$k timeout -k 10 300 python -u bench.py --batch-size 400 --steps 30 --warmup 8 > gpurun_out/bench_shallow_$k.log 2>&1 || { tail -n 20 gpurun_out/bench_shallow_$k.log; exit 2; }
echo "shallow_k $k $(tail -n 1 gpurun_out/bench_shallow_$k.log | grep -o '"value": [0-9.]*')"
done
