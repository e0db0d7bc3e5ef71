#!/bin/bash
# GPU tests + smoke + per-layer conv sweep + stock comparator at batch 256.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --tb=line --timeout 120 --timeout-method thread > gpurun_out/pytest_gpu.log 2>&1
rc=$?; tail -n 5 gpurun_out/pytest_gpu.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python -u -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1 || { tail -n 20 gpurun_out/smoke.log; exit 1; }
tail -n 2 gpurun_out/smoke.log
timeout -k 10 300 python -u benchmarks/conv_bench.py --batch 128 --cfgs 0,1,2,3 --wcfgs 0,1,2 --nsplits 0 --out gpurun_out/conv_bench_b128.csv > gpurun_out/conv_bench_b128.log 2>&1 || { tail -n 20 gpurun_out/conv_bench_b128.log; exit 1; }
tail -n 1 gpurun_out/conv_bench_b128.log
timeout -k 10 300 python -u bench.py --engine stock --channels-last --batch-size 256 --steps 20 --warmup 5 > gpurun_out/bench_stock_cl_b256.log 2>&1 || { tail -n 20 gpurun_out/bench_stock_cl_b256.log; exit 1; }
tail -n 1 gpurun_out/bench_stock_cl_b256.log
timeout -k 10 300 python -u bench.py --steps 30 --warmup 8 > gpurun_out/bench_native_default.log 2>&1 || { tail -n 20 gpurun_out/bench_native_default.log; exit 1; }
tail -n 1 gpurun_out/bench_native_default.log
