#!/bin/bash
# ReLU masks as sign bits: numerics, then A/B of the default benchmark
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests/test_kernels_gpu.py -x -q --timeout 300 --timeout-method thread -k "sign_bits or relu_backward_fusion or dgrad_mask or resnet_fwd_bwd or deferred or bench_configuration" > gpurun_out/mb_pytest.log 2>&1; rc=$?
tail -8 gpurun_out/mb_pytest.log
[ $rc -eq 0 ] || exit $rc
for v in 0 1; do
  PDA_RELU_MASK_BITS=$v timeout -k 10 300 python -u bench.py --steps 20 --warmup 10 > gpurun_out/mb_bench_$v.log 2>&1 || { tail -20 gpurun_out/mb_bench_$v.log; exit 1; }
  echo "PDA_RELU_MASK_BITS=$v: $(tail -1 gpurun_out/mb_bench_$v.log | cut -c1-230)"
done
