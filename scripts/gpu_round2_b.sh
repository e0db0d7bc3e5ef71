#!/bin/bash
# GPU session B: full GPU test-suite, smoke, single-rank DDP bench (native comm, fused wire), plain bench
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/b_pytest.log 2>&1; rc=$?
tail -15 gpurun_out/b_pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python -u __graft_entry__.py smoke > gpurun_out/b_smoke.log 2>&1; rc=$?; tail -3 gpurun_out/b_smoke.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -u bench.py --steps 20 --warmup 10 --force-ddp > gpurun_out/b_bench_ddp.log 2>&1; rc=$?; tail -2 gpurun_out/b_bench_ddp.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -u bench.py --steps 20 --warmup 10 --force-ddp --comm torch > gpurun_out/b_bench_ddp_torch.log 2>&1; rc=$?; tail -1 gpurun_out/b_bench_ddp_torch.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -u bench.py --steps 20 --warmup 10 > gpurun_out/b_bench_plain.log 2>&1; rc=$?; tail -1 gpurun_out/b_bench_plain.log
exit $rc
