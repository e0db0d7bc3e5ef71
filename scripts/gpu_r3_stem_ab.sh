#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_kernels_gpu.py -k "stem or resnet_fwd_bwd" -x -q --timeout 200 --timeout-method thread 2>&1 | tail -n 5 || exit 1
for rep in 1 2; do for m in 1 0; do
  PDA_FUSE_STEM_BWD=$m timeout -k 10 300 python -u bench.py --steps 30 --warmup 10 --no-compare-stock > gpurun_out/stem_ab_$m.log 2>&1 || { tail -n 20 gpurun_out/stem_ab_$m.log; exit 1; }
  echo "PDA_FUSE_STEM_BWD=$m $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/stem_ab_$m.log)" | tee -a gpurun_out/stem_ab.txt
done; done
