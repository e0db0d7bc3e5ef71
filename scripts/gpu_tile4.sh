#!/bin/bash
# correctness of the 256x128 tile, then a per-layer sweep of it against the auto choice
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_kernels_gpu.py -m gpu -x -q --tb=short --timeout 120 --timeout-method thread -k "conv" > gpurun_out/pytest_tile4.log 2>&1 || { tail -n 25 gpurun_out/pytest_tile4.log | cut -c1-300; exit 1; }
tail -n 3 gpurun_out/pytest_tile4.log
timeout -k 10 200 python -u benchmarks/conv_bench.py --batch 128 --cfgs 0,4 --only fd --out gpurun_out/conv_tile4_b128.csv > gpurun_out/conv_tile4_b128.txt 2>&1 || { tail -n 20 gpurun_out/conv_tile4_b128.txt; exit 2; }
timeout -k 10 200 python -u benchmarks/conv_bench.py --batch 256 --cfgs 0,4 --only fd --out gpurun_out/conv_tile4_b256.csv > gpurun_out/conv_tile4_b256.txt 2>&1 || { tail -n 20 gpurun_out/conv_tile4_b256.txt; exit 3; }
tail -n 8 gpurun_out/conv_tile4_b256.txt
