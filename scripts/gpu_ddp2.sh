#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_kernels_gpu.py -m gpu -x -q --tb=short --timeout 120 --timeout-method thread -k "ddp or rccl" > gpurun_out/pytest_gpu_ddp.log 2>&1
rc=$?; tail -n 30 gpurun_out/pytest_gpu_ddp.log; [ $rc -ne 0 ] && exit $rc
for extra in "" "--force-ddp" "--force-ddp --native-comm"; do
timeout -k 10 300 python -u bench.py --batch-size 256 --steps 30 --warmup 8 $extra > gpurun_out/bench_ddp_tmp.log 2>&1 || { tail -n 30 gpurun_out/bench_ddp_tmp.log; exit 1; }
echo "b256 [$extra] $(tail -n 1 gpurun_out/bench_ddp_tmp.log | grep -o '"value": [0-9.]*')"
done
