#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -u benchmarks/dgrad_fusion_bench.py --batch 256 --detail > gpurun_out/dgrad_fusion_b256.txt 2>&1 || { tail -n 20 gpurun_out/dgrad_fusion_b256.txt; exit 1; }
cat gpurun_out/dgrad_fusion_b256.txt
