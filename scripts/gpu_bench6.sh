#!/bin/bash
# GPU tests + dgrad/wgrad sweep + CPU-overhead probe (tiny batch) + benches + kernel trace.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --tb=short --timeout 120 --timeout-method thread > gpurun_out/pytest_gpu.log 2>&1
rc=$?; tail -n 15 gpurun_out/pytest_gpu.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python -u benchmarks/conv_bench.py --batch 128 --only dw --cfgs 0,16 --wcfgs 0 --out gpurun_out/conv_bench_b128_v2.csv > gpurun_out/conv_bench_b128_v2.log 2>&1 || { tail -n 20 gpurun_out/conv_bench_b128_v2.log; exit 1; }
tail -n 1 gpurun_out/conv_bench_b128_v2.log
timeout -k 10 300 python -u bench.py --batch-size 8 --steps 30 --warmup 8 > gpurun_out/bench_native_b8.log 2>&1 || { tail -n 20 gpurun_out/bench_native_b8.log; exit 1; }
tail -n 1 gpurun_out/bench_native_b8.log | cut -c1-200
timeout -k 10 300 python -u bench.py --batch-size 128 --steps 30 --warmup 8 > gpurun_out/bench_native_b128.log 2>&1 || { tail -n 20 gpurun_out/bench_native_b128.log; exit 1; }
tail -n 1 gpurun_out/bench_native_b128.log | cut -c1-200
timeout -k 10 300 python -u bench.py --steps 30 --warmup 8 > gpurun_out/bench_native_default.log 2>&1 || { tail -n 20 gpurun_out/bench_native_default.log; exit 1; }
tail -n 1 gpurun_out/bench_native_default.log | cut -c1-200
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_native4 -o r -- python3 $GRAFT_REPO_ROOT/bench.py --batch-size 128 --steps 5 --warmup 3 > $GRAFT_REPO_ROOT/gpurun_out/prof_native4.log 2>&1 || { tail -n 20 $GRAFT_REPO_ROOT/gpurun_out/prof_native4.log; exit 1; }
echo profile done
