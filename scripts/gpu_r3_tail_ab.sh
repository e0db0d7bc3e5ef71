#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for rep in 1 2; do for m in 2 1 0; do
  PDA_FUSE_TAIL_BWD=$m timeout -k 10 300 python -u bench.py --steps 30 --warmup 10 --no-compare-stock > gpurun_out/tail_ab_$m.log 2>&1 || { tail -n 20 gpurun_out/tail_ab_$m.log; exit 1; }
  echo "PDA_FUSE_TAIL_BWD=$m $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/tail_ab_$m.log)" | tee -a gpurun_out/tail_ab.txt
done; done
