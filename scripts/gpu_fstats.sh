#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -u benchmarks/conv_bench.py --batch 256 --only f --out gpurun_out/f_stats.csv > gpurun_out/f_stats.txt 2>&1 || exit 1
timeout -k 10 300 python -u benchmarks/conv_bench.py --batch 256 --only f --no-stats --out gpurun_out/f_nostats.csv > gpurun_out/f_nostats.txt 2>&1 || exit 1
tail -n 1 gpurun_out/f_stats.txt; tail -n 1 gpurun_out/f_nostats.txt
