#!/bin/bash
# whole-step hipGraph replay with / without the side-stream branch recorded into the graph
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_kernels_gpu.py -x -q --timeout 200 --timeout-method thread -k "graphed" > gpurun_out/g3_pytest.log 2>&1; rc=$?
tail -n 3 gpurun_out/g3_pytest.log
[ $rc -eq 0 ] || exit $rc
run() { timeout -k 10 300 python -u bench.py --steps 20 --warmup 10 $2 > gpurun_out/g3.log 2>&1 || { tail -n 20 gpurun_out/g3.log; exit 1; }; echo "$1: $(tail -n 1 gpurun_out/g3.log | grep -o '"value": [0-9.]*, "unit": "images/sec", "n_gpus": 1, "steps": 20, "warmup": 10, "ms_per_step": [0-9.]*') $(tail -n 1 gpurun_out/g3.log | grep -o '"hip_graph": [a-z]*')"; }
run "eager" ""
run "graph, two branches" "--graph"
PDA_WGRAD_STREAM_IN_GRAPH=0 run "graph, one stream" "--graph"
run "eager" ""
