#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "sums or fused or link or relu_backward or deferred or many_tiles or bottleneck or block" --timeout 200 --timeout-method thread > gpurun_out/slots_tests.log 2>&1 || { tail -n 40 gpurun_out/slots_tests.log | cut -c1-300; exit 1; }
tail -n 2 gpurun_out/slots_tests.log
for rep in 1 2; do
  for s in 1 0; do
    PDA_BN_SUMS_SLOTS=$s timeout -k 10 300 python -u bench.py --steps 20 --warmup 10 > gpurun_out/slots_$s.log 2>&1 || { tail -n 20 gpurun_out/slots_$s.log; exit 2; }
    echo "PDA_BN_SUMS_SLOTS=$s: $(tail -n 1 gpurun_out/slots_$s.log | grep -o '"ms_per_step": [0-9.]*')"
  done
done
