#!/bin/bash
# GPU session E: multi-rank-on-one-GPU tests (O2 twin, hvd allgather / Adasum), whole GPU suite, bench vs stock
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 600 python -u -m pytest tests/test_p2p_gpu.py -x -q --timeout 200 --timeout-method thread > gpurun_out/e_pytest_p2p.log 2>&1; rc=$?
tail -25 gpurun_out/e_pytest_p2p.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread --deselect tests/test_p2p_gpu.py > gpurun_out/e_pytest_all.log 2>&1; rc=$?
tail -8 gpurun_out/e_pytest_all.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -u bench.py --steps 20 --warmup 10 --compare-stock > gpurun_out/e_bench_stock.log 2>&1; rc=$?; tail -1 gpurun_out/e_bench_stock.log | cut -c1-600
exit $rc
