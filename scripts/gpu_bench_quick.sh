#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -u bench.py --steps 20 --warmup 10 > gpurun_out/q_bench.log 2>&1; rc=$?; tail -1 gpurun_out/q_bench.log | cut -c1-300
exit $rc
