#!/bin/bash
# GPU session D: new kernels (linear, eval fold, peer, adasum, O2 fused wire multi-rank) then the whole GPU suite
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 600 python -u -m pytest tests/test_kernels_gpu.py -x -q --timeout 200 --timeout-method thread -k "linear_runs or eval_batchnorm or peer_broadcast or bench_configuration or ddp_single_rank or native_rccl" > gpurun_out/d_pytest_new.log 2>&1; rc=$?
tail -25 gpurun_out/d_pytest_new.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python -u -m pytest tests/test_p2p_gpu.py -x -q --timeout 200 --timeout-method thread > gpurun_out/d_pytest_p2p.log 2>&1; rc=$?
tail -25 gpurun_out/d_pytest_p2p.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread --deselect tests/test_p2p_gpu.py > gpurun_out/d_pytest_all.log 2>&1; rc=$?
tail -8 gpurun_out/d_pytest_all.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -u bench.py --steps 20 --warmup 10 --compare-stock > gpurun_out/d_bench_stock.log 2>&1; rc=$?; tail -1 gpurun_out/d_bench_stock.log | cut -c1-400
exit $rc
