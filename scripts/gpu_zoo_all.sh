#!/bin/bash
# model-zoo GPU tests, then the per-family throughput table
set -o pipefail
bash scripts/gpu_zoo.sh > gpurun_out/zoo_tests_tail.txt 2>&1 || { tail -n 40 gpurun_out/zoo_tests_tail.txt; exit 1; }
tail -n 2 gpurun_out/zoo_tests_tail.txt
bash scripts/gpu_zoo_bench.sh
