#!/bin/bash
# ring wgrad: numerics, then per-layer A/B against the register-staged kernel at the bench batch
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_kernels_gpu.py -k "wgrad" -x -q --timeout 120 --timeout-method thread 2>&1 | tail -n 15 || exit 1
timeout -k 10 500 python -u benchmarks/conv_bench.py --batch 400 --only w --iters 10 --wcfgs 0,1,12,13,65,68,69,70 --out gpurun_out/wring_b400.csv > gpurun_out/wring_b400.txt 2>&1 || { tail -n 20 gpurun_out/wring_b400.txt; exit 1; }
tail -n 3 gpurun_out/wring_b400.txt
