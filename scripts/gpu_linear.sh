#!/bin/bash
# Linear layers: this package's kernels vs the library GEMM, plus the Linear numerics tests
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "linear or Linear or split_k" --timeout 120 --timeout-method thread > gpurun_out/linear_tests.log 2>&1 || { tail -n 30 gpurun_out/linear_tests.log | cut -c1-300; exit 1; }
tail -n 2 gpurun_out/linear_tests.log
timeout -k 10 200 python -u benchmarks/linear_bench.py --batch 128 --direct-grad > gpurun_out/linear_bench.log 2>&1 || { tail -n 20 gpurun_out/linear_bench.log; exit 2; }
cat gpurun_out/linear_bench.log
PDA_LINEAR_SPLITK_MAX_BATCH=0 PDA_LINEAR_SPLITK_WORKGROUPS=0 timeout -k 10 200 python -u benchmarks/linear_bench.py --batch 128 --direct-grad > gpurun_out/linear_bench_nosplit.log 2>&1 || { tail -n 20 gpurun_out/linear_bench_nosplit.log; exit 3; }
cat gpurun_out/linear_bench_nosplit.log
PDA_LINEAR_SPLITK_WORKGROUPS=512 timeout -k 10 200 python -u benchmarks/linear_bench.py --batch 128 --direct-grad > gpurun_out/linear_bench_512.log 2>&1 || { tail -n 20 gpurun_out/linear_bench_512.log; exit 4; }
cat gpurun_out/linear_bench_512.log
