#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "clipped_relu or batchnorm or bn_" --timeout 120 --timeout-method thread > gpurun_out/relu6_tests.log 2>&1 || { tail -n 40 gpurun_out/relu6_tests.log | cut -c1-300; exit 1; }
tail -n 2 gpurun_out/relu6_tests.log
timeout -k 10 300 python -u -m pytest tests/test_model_zoo_gpu.py -m gpu -x -q -k "mobilenet" --timeout 200 --timeout-method thread > gpurun_out/relu6_zoo.log 2>&1 || { tail -n 40 gpurun_out/relu6_zoo.log | cut -c1-300; exit 2; }
tail -n 2 gpurun_out/relu6_zoo.log
timeout -k 10 300 python -u benchmarks/zoo_bench.py --archs mobilenet_v2,mnasnet1_0 > gpurun_out/zoo_relu6.log 2>&1 || { tail -n 20 gpurun_out/zoo_relu6.log; exit 3; }
grep "img/s" gpurun_out/zoo_relu6.log
