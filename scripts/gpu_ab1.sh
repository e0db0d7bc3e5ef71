#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_kernels_gpu.py -m gpu -x -q --tb=short --timeout 120 --timeout-method thread -k "conv or dgrad or wgrad or resnet" > gpurun_out/pytest_gpu_conv.log 2>&1
rc=$?; tail -n 5 gpurun_out/pytest_gpu_conv.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python -u benchmarks/conv_bench.py --batch 128 --only d --cfgs 0,16 --out gpurun_out/conv_bench_b128_v5.csv > gpurun_out/conv_bench_b128_v5.log 2>&1 || { tail -n 20 gpurun_out/conv_bench_b128_v5.log; exit 1; }
grep "'s': 2" gpurun_out/conv_bench_b128_v5.log | sed -E "s/.*'H': ([0-9]+), 'Cin': ([0-9]+), 'Cout': ([0-9]+), 'k': ([0-9]+).*'dgrad_all': '([^']*)'.*/\1 \2 \3 k\4 \5/"
tail -n 1 gpurun_out/conv_bench_b128_v5.log
for rep in 1 2; do
for s in 1 0; do
PDA_STAT_SLOTS=$s timeout -k 10 300 python -u bench.py --batch-size 128 --steps 30 --warmup 8 > gpurun_out/bench_ab_slots$s.log 2>&1 || { tail -n 20 gpurun_out/bench_ab_slots$s.log; exit 1; }
echo "slots=$s $(tail -n 1 gpurun_out/bench_ab_slots$s.log | cut -c95-125)"
done
done
PDA_STAT_SLOTS=1 timeout -k 10 300 python -u bench.py --batch-size 256 --steps 30 --warmup 8 | cut -c95-125
PDA_STAT_SLOTS=0 timeout -k 10 300 python -u bench.py --batch-size 256 --steps 30 --warmup 8 | cut -c95-125
