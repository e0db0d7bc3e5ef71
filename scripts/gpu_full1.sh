#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --tb=short --timeout 120 --timeout-method thread > gpurun_out/pytest_gpu.log 2>&1
rc=$?; tail -n 4 gpurun_out/pytest_gpu.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python -u -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1 || { tail -n 20 gpurun_out/smoke.log; exit 1; }
timeout -k 10 400 python -u benchmarks/conv_bench.py --batch 128 --cfgs 0 --wcfgs 0 --out gpurun_out/conv_bench_b128_final.csv > gpurun_out/conv_bench_b128_final.log 2>&1 || { tail -n 20 gpurun_out/conv_bench_b128_final.log; exit 1; }
tail -n 1 gpurun_out/conv_bench_b128_final.log
for b in 128 256 512; do
timeout -k 10 300 python -u bench.py --batch-size $b --steps 30 --warmup 8 > gpurun_out/bench_native_b$b.log 2>&1 || { tail -n 20 gpurun_out/bench_native_b$b.log; exit 1; }
echo "native b$b $(tail -n 1 gpurun_out/bench_native_b$b.log | cut -c95-125) $(tail -n 1 gpurun_out/bench_native_b$b.log | grep -o '"ms_per_step": [0-9.]*')"
done
for b in 128 256; do
timeout -k 10 300 python -u bench.py --engine stock --channels-last --batch-size $b --steps 20 --warmup 5 > gpurun_out/bench_stock_cl_b$b.log 2>&1 || { tail -n 20 gpurun_out/bench_stock_cl_b$b.log; exit 1; }
echo "stock  b$b $(tail -n 1 gpurun_out/bench_stock_cl_b$b.log | cut -c95-125)"
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_native5 -o r -- python3 $R/bench.py --batch-size 256 --steps 5 --warmup 3 > $R/gpurun_out/prof_native5.log 2>&1 || { tail -n 20 $R/gpurun_out/prof_native5.log; exit 1; }
echo profile done
