#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python -u benchmarks/conv_bench.py --batch 256 --cfgs 0 --only fd --out gpurun_out/conv_sweep_b256.csv > gpurun_out/conv_sweep_b256.txt 2>&1 || { tail -n 20 gpurun_out/conv_sweep_b256.txt; exit 1; }
tail -n 1 gpurun_out/conv_sweep_b256.txt
