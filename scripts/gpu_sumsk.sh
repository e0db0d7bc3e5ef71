#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for k in 0 256 600 1100 2200 100000000; do
PDA_FUSE_BN_SUMS_MIN_K=$k timeout -k 10 300 python -u bench.py --batch-size 400 --steps 30 --warmup 8 > gpurun_out/bench_sumsk_$k.log 2>&1 || { tail -n 20 gpurun_out/bench_sumsk_$k.log; exit 2; }
echo "min_k $k $(tail -n 1 gpurun_out/bench_sumsk_$k.log | grep -o '"value": [0-9.]*')"
done
