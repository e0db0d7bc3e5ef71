#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "depthwise or pool or dw_" --timeout 200 --timeout-method thread > gpurun_out/dw2_tests.log 2>&1 || { tail -n 40 gpurun_out/dw2_tests.log | cut -c1-300; exit 1; }
tail -n 2 gpurun_out/dw2_tests.log
timeout -k 10 400 python -u -m pytest tests/test_model_zoo_gpu.py -m gpu -x -q -k "shufflenet or mobilenet or mnasnet" --timeout 200 --timeout-method thread > gpurun_out/dw2_zoo.log 2>&1 || { tail -n 40 gpurun_out/dw2_zoo.log | cut -c1-300; exit 2; }
tail -n 2 gpurun_out/dw2_zoo.log
timeout -k 10 300 python -u benchmarks/zoo_bench.py --archs shufflenet_v2_x1_0,mobilenet_v2 --steps 30 > gpurun_out/zoo_dw2.log 2>&1 || { tail -n 20 gpurun_out/zoo_dw2.log; exit 3; }
grep "img/s" gpurun_out/zoo_dw2.log
timeout -k 10 300 python -u benchmarks/zoo_bench.py --archs shufflenet_v2_x1_0 --steps 30 --graph > gpurun_out/zoo_dw2g.log 2>&1 || { tail -n 20 gpurun_out/zoo_dw2g.log; exit 3; }
grep "img/s" gpurun_out/zoo_dw2g.log
