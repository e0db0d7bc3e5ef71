#!/bin/bash
# which inner hand-offs should skip their normalisation pass: bn1 -> 3x3 conv2 (bit 0), bn2 -> 1x1 conv3 (bit 1)
set -o pipefail
mkdir -p gpurun_out
for rep in 1 2; do
for m in 3 2 1 0; do
  PDA_DEFER_INNER_MASK=$m timeout -k 10 300 python -u bench.py --steps 20 --warmup 10 > gpurun_out/dm.log 2>&1 || { tail -n 20 gpurun_out/dm.log; exit 1; }
  echo "mask $m: $(tail -n 1 gpurun_out/dm.log | grep -o '"ms_per_step": [0-9.]*')"
done
done
