#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python -u benchmarks/conv_bench.py --batch 400 --wcfgs 0,12,13 --only w --out gpurun_out/wgrad_b400.csv > gpurun_out/wgrad_b400.txt 2>&1 || { tail -n 20 gpurun_out/wgrad_b400.txt; exit 1; }
tail -n 1 gpurun_out/wgrad_b400.txt
