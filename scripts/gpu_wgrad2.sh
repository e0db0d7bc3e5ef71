#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python -u benchmarks/conv_bench.py --batch 400 --wcfgs 0 --wblocks 256,384,512,768,1024 --only w --out gpurun_out/wgrad_blocks_b400.csv > gpurun_out/wgrad_blocks_b400.txt 2>&1 || { tail -n 20 gpurun_out/wgrad_blocks_b400.txt; exit 1; }
timeout -k 10 500 python -u benchmarks/conv_bench.py --batch 400 --wcfgs 0 --only w --out gpurun_out/wgrad_auto_b400.csv > gpurun_out/wgrad_auto_b400.txt 2>&1 || exit 1
tail -n 1 gpurun_out/wgrad_auto_b400.txt
