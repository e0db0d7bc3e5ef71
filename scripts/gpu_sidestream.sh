#!/bin/bash
# wgrad on the backward side stream: A/B of the default benchmark, then the fused-block / DDP numerics tests
set -o pipefail
mkdir -p gpurun_out
for v in 0 1; do
  PDA_WGRAD_STREAM=$v timeout -k 10 300 python -u bench.py --steps 20 --warmup 10 > gpurun_out/ss_bench_$v.log 2>&1 || { tail -20 gpurun_out/ss_bench_$v.log; exit 1; }
  echo "PDA_WGRAD_STREAM=$v: $(tail -1 gpurun_out/ss_bench_$v.log | cut -c1-230)"
done
PDA_WGRAD_STREAM=1 timeout -k 10 300 python -u bench.py --steps 20 --warmup 10 --force-ddp > gpurun_out/ss_bench_ddp.log 2>&1 || { tail -20 gpurun_out/ss_bench_ddp.log; exit 1; }
echo "ddp: $(tail -1 gpurun_out/ss_bench_ddp.log | cut -c1-230)"
timeout -k 10 900 python -u -m pytest tests/test_kernels_gpu.py -x -q --timeout 200 --timeout-method thread -k "resnet or ddp or fused or block" > gpurun_out/ss_pytest.log 2>&1; rc=$?
tail -5 gpurun_out/ss_pytest.log
exit $rc
