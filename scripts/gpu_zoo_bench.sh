#!/bin/bash
# training throughput of every model family (one GPU, fp16 O2)
set -o pipefail
mkdir -p gpurun_out; rm -f gpurun_out/zoo_bench.txt
timeout -k 10 900 python -u benchmarks/zoo_bench.py --batch 128 --out gpurun_out/zoo_bench.txt 2>&1 | tail -n 30
