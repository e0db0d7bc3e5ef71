#!/bin/bash
# headline bench with the ring wgrad among the tuner's candidates, for each choice of normalise-on-load hand-offs
set -o pipefail
mkdir -p gpurun_out
for m in 3 2 0; do
  PDA_DEFER_INNER_MASK=$m timeout -k 10 300 python -u bench.py --steps 30 --warmup 10 --no-compare-stock > gpurun_out/wring_bench_m$m.log 2>&1 || { tail -n 20 gpurun_out/wring_bench_m$m.log; exit 1; }
  echo "mask=$m $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/wring_bench_m$m.log) $(grep -o '"value": [0-9.]*' gpurun_out/wring_bench_m$m.log)" | tee -a gpurun_out/wring_bench_summary.txt
done
