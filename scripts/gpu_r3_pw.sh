#!/bin/bash
# A-stationary pointwise fprop: numerics, then per-layer A/B against the tiled kernels at the bench batch
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_kernels_gpu.py -k "pointwise_a_stationary" -x -q --timeout 120 --timeout-method thread 2>&1 | tail -n 25 || exit 1
timeout -k 10 500 python -u benchmarks/conv_bench.py --batch 400 --only f --iters 10 --cfgs 0,1,2,9,12,33 > gpurun_out/pw_b400.txt 2>&1 || { tail -n 20 gpurun_out/pw_b400.txt; exit 1; }
tail -n 3 gpurun_out/pw_b400.txt
