#!/bin/bash
# effect of the Linear split-K paths on the models with big fully connected layers, and the headline bench (the
# conv kernel gained the split-K branch)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -u benchmarks/zoo_bench.py --archs alexnet,vgg16,vgg16_bn > gpurun_out/zoo_linear_new.log 2>&1 || { tail -n 20 gpurun_out/zoo_linear_new.log; exit 1; }
grep "img/s" gpurun_out/zoo_linear_new.log
PDA_LINEAR_SPLITK_MAX_BATCH=0 PDA_LINEAR_SPLITK_WORKGROUPS=0 timeout -k 10 300 python -u benchmarks/zoo_bench.py --archs alexnet,vgg16 > gpurun_out/zoo_linear_old.log 2>&1 || { tail -n 20 gpurun_out/zoo_linear_old.log; exit 2; }
grep "img/s" gpurun_out/zoo_linear_old.log
timeout -k 10 300 python -u bench.py > gpurun_out/bench_after_splitk.log 2>&1 || { tail -n 20 gpurun_out/bench_after_splitk.log; exit 3; }
tail -n 1 gpurun_out/bench_after_splitk.log | cut -c1-200
