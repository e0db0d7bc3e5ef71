#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_kernels_gpu.py -k "wgrad or normalised" -x -q --timeout 120 --timeout-method thread 2>&1 | tail -n 8 || exit 1
timeout -k 10 300 python -u benchmarks/conv_bench.py --batch 400 --only w --iters 10 --wcfgs 0,65,66,68,69 > gpurun_out/wring66_b400.txt 2>&1 || { tail -n 20 gpurun_out/wring66_b400.txt; exit 1; }
grep "'Cout': 64," gpurun_out/wring66_b400.txt | cut -c1-60,200-400
bash scripts/gpu_r3_wring_bench.sh
