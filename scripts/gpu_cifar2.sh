#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout -k 10 300 python -u bench.py --engine stock --arch resnet18 --image-size 32 --batch-size 800 --precision fp16 --steps 40 --warmup 10 > gpurun_out/bench_r18c_stock_nchw_fp16.log 2>&1
echo "r18 cifar b800 stock nchw fp16 $(tail -n 1 gpurun_out/bench_r18c_stock_nchw_fp16.log | cut -c95-125)"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r18c_native_fp32 -o r -- python3 $R/bench.py --arch resnet18 --image-size 32 --batch-size 800 --precision fp32 --steps 10 --warmup 5 > $R/gpurun_out/prof_r18c_native_fp32.log 2>&1 || { tail -n 20 $R/gpurun_out/prof_r18c_native_fp32.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r18c_stock_fp32 -o r -- python3 $R/bench.py --engine stock --arch resnet18 --image-size 32 --batch-size 800 --precision fp32 --steps 10 --warmup 5 > $R/gpurun_out/prof_r18c_stock_fp32.log 2>&1 || { tail -n 20 $R/gpurun_out/prof_r18c_stock_fp32.log; exit 1; }
find $R/gpurun_out -name "*kernel_trace.csv" -delete; echo done
