#!/bin/bash
# the launch-bound configuration (ResNet-18, 32x32 input, fp32 and fp16): does the side stream cost host time there?
set -o pipefail
mkdir -p gpurun_out
run() { timeout -k 10 200 python -u bench.py --arch resnet18 --image-size 32 --batch-size 400 --steps 40 --warmup 15 $2 > gpurun_out/cs.log 2>&1 || { tail -n 20 gpurun_out/cs.log; exit 1; }; echo "$1 $2: $(tail -n 1 gpurun_out/cs.log | grep -o '"value": [0-9.]*') $(tail -n 1 gpurun_out/cs.log | grep -o '"ms_per_step": [0-9.]*') $(tail -n 1 gpurun_out/cs.log | grep -o '"host_issue_ms_per_step": [0-9.]*')"; }
for p in "--precision fp32" "--precision fp16"; do
  PDA_WGRAD_STREAM=1 run "side stream on " "$p"
  PDA_WGRAD_STREAM=0 run "side stream off" "$p"
  PDA_WGRAD_STREAM=1 run "side stream on , graph" "$p --graph"
  PDA_WGRAD_STREAM=0 run "side stream off, graph" "$p --graph"
done
