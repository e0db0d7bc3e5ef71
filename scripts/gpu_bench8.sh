#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_kernels_gpu.py -m gpu -x -q --tb=short --timeout 120 --timeout-method thread -k "conv or dgrad or wgrad" > gpurun_out/pytest_gpu_conv.log 2>&1
rc=$?; tail -n 5 gpurun_out/pytest_gpu_conv.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python -u benchmarks/conv_bench.py --batch 128 --only d --cfgs 0,16 --out gpurun_out/conv_bench_b128_v4.csv > gpurun_out/conv_bench_b128_v4.log 2>&1 || { tail -n 20 gpurun_out/conv_bench_b128_v4.log; exit 1; }
grep "'s': 2" gpurun_out/conv_bench_b128_v4.log | sed -E "s/.*'H': ([0-9]+), 'Cin': ([0-9]+), 'Cout': ([0-9]+), 'k': ([0-9]+).*'dgrad_all': '([^']*)'.*/\1 \2 \3 k\4 \5/"
for b in 256 512 640; do
timeout -k 10 300 python -u bench.py --batch-size $b --steps 20 --warmup 6 > gpurun_out/bench_native_b$b.log 2>&1 || { tail -n 20 gpurun_out/bench_native_b$b.log; exit 1; }
tail -n 1 gpurun_out/bench_native_b$b.log | cut -c1-200
done
timeout -k 10 300 python -u bench.py --engine stock --channels-last --batch-size 512 --steps 12 --warmup 4 > gpurun_out/bench_stock_cl_b512.log 2>&1 || { tail -n 20 gpurun_out/bench_stock_cl_b512.log; exit 1; }
tail -n 1 gpurun_out/bench_stock_cl_b512.log | cut -c1-200
