#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests/test_kernels_gpu.py -m gpu -x -q --tb=short --timeout 120 --timeout-method thread > gpurun_out/pytest_gpu.log 2>&1
rc=$?; tail -n 4 gpurun_out/pytest_gpu.log; [ $rc -ne 0 ] && exit $rc
for b in 128 256; do
timeout -k 10 300 python -u bench.py --batch-size $b --steps 30 --warmup 8 > gpurun_out/bench_native_b$b.log 2>&1 || { tail -n 20 gpurun_out/bench_native_b$b.log; exit 1; }
echo "native b$b $(tail -n 1 gpurun_out/bench_native_b$b.log | grep -o '"value": [0-9.]*')"
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_native7 -o r -- python3 $R/bench.py --batch-size 256 --steps 5 --warmup 3 > $R/gpurun_out/prof_native7.log 2>&1 || { tail -n 20 $R/gpurun_out/prof_native7.log; exit 1; }
find $R/gpurun_out -name "*kernel_trace.csv" -delete
echo done
