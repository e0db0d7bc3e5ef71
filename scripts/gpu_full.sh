#!/bin/bash
# everything the round-end driver runs on the GPU box: all gpu-marked tests, smoke(), default bench (+ the stock
# comparator in the same process)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --tb=short --timeout 300 --timeout-method thread > gpurun_out/pytest_gpu_all.log 2>&1 || { tail -n 40 gpurun_out/pytest_gpu_all.log | cut -c1-400; exit 1; }
tail -n 3 gpurun_out/pytest_gpu_all.log
timeout -k 10 300 python -u -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1 || { tail -n 20 gpurun_out/smoke.log; exit 2; }
tail -n 1 gpurun_out/smoke.log
timeout -k 10 300 python -u bench.py > gpurun_out/bench_default.log 2>&1 || { tail -n 20 gpurun_out/bench_default.log; exit 3; }
tail -n 1 gpurun_out/bench_default.log
timeout -k 10 400 python -u bench.py --steps 20 --warmup 10 --compare-stock > gpurun_out/bench_vs_stock.log 2>&1 || { tail -n 20 gpurun_out/bench_vs_stock.log; exit 4; }
tail -n 1 gpurun_out/bench_vs_stock.log | cut -c1-420
