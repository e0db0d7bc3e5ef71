#!/bin/bash
# DDP bucket launches from the backward side stream: single-rank RCCL tests, multi-rank-on-one-GPU tests, --force-ddp bench
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 600 python -u -m pytest tests/test_kernels_gpu.py -x -q --timeout 300 --timeout-method thread -k "ddp or rccl or graphed" > gpurun_out/ds_pytest.log 2>&1; rc=$?
tail -4 gpurun_out/ds_pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python -u -m pytest tests/test_p2p_gpu.py -x -q --timeout 300 --timeout-method thread > gpurun_out/ds_pytest_p2p.log 2>&1; rc=$?
tail -4 gpurun_out/ds_pytest_p2p.log
[ $rc -eq 0 ] || exit $rc
for extra in "" "--force-ddp" "--force-ddp --comm torch"; do
  timeout -k 10 300 python -u bench.py --steps 20 --warmup 10 $extra > gpurun_out/ds_bench.log 2>&1 || { tail -20 gpurun_out/ds_bench.log; exit 1; }
  echo "bench $extra: $(tail -1 gpurun_out/ds_bench.log | cut -c95-230)"
done
