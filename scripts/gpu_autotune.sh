#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for i in 1 2; do
for f in "--no-autotune" "--autotune"; do
timeout -k 10 400 python -u bench.py --steps 30 --warmup 10 $f > gpurun_out/bench_at$f.log 2>&1 || { tail -n 20 gpurun_out/bench_at$f.log; exit 2; }
echo "$f $(tail -n 1 gpurun_out/bench_at$f.log | grep -o '"value": [0-9.]*')"
done
done
