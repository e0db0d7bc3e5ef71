#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_kernels_gpu.py -k "lds_dma_ring_tiles" -x -q --timeout 120 --timeout-method thread 2>&1 | tail -n 8 || exit 1
timeout -k 10 500 python -u benchmarks/conv_bench.py --batch 400 --only fd --iters 10 --cfgs 0,1,12,5,6,7,13 > gpurun_out/ring4_b400.txt 2>&1 || { tail -n 20 gpurun_out/ring4_b400.txt; exit 1; }
tail -n 2 gpurun_out/ring4_b400.txt
